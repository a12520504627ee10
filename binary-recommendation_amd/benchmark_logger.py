"""Utilisation sampler of the reference's training runs — src/origin_models/svd/benchmarkLogger.py:9-40: a daemon thread that
appends [time, CPU %, process RSS MB, GPU %] rows to a CSV every `pollTime` seconds while `active` is truthy (started around
`fit`: trainers/twoTower.py:202-223).  The reference reads the GPU load through GPUtil (nvidia-smi); here the MI355X's
`gpu_busy_percent` comes from the amdgpu sysfs node (no subprocess per sample), with `rocm-smi --showuse` as a fallback and 0
when neither exists.  The matplotlib figure the reference draws on exit (benchmarkLogger.py:41-80) is optional: `plot=True`
needs matplotlib."""
from __future__ import annotations

import csv
import glob
import os
import re
import subprocess
import threading
import time

HEADER = ["Time (s)", "CPU %", "Memory MB", "GPU %"]


def _sysfs_busy_nodes():
    return sorted(glob.glob("/sys/class/drm/card*/device/gpu_busy_percent"))


def gpu_busy_percent(index: int = 0) -> float:
    """GPU `index` load in percent: amdgpu sysfs, else rocm-smi, else 0.0 (the reference reports 0 without a GPU)."""
    nodes = _sysfs_busy_nodes()
    if index < len(nodes):
        try:
            with open(nodes[index]) as f:
                return float(f.read().strip())
        except (OSError, ValueError):
            pass
    try:
        out = subprocess.run(["rocm-smi", "--showuse"], capture_output=True, text=True, timeout=5).stdout
        m = re.findall(r"GPU\[(\d+)\].*?GPU use \(%\):\s*([\d.]+)", out)
        for g, v in m:
            if int(g) == index:
                return float(v)
    except (OSError, subprocess.SubprocessError):
        pass
    return 0.0


class benchThread(threading.Thread):
    """benchThread(pollTime, active, filepath) as in benchmarkLogger.py:9-27; set `.active = 0` and join() to stop."""

    def __init__(self, pollTime, active, filepath, gpu_index: int = 0, plot: bool = False):
        threading.Thread.__init__(self)
        self.pollTime, self.active, self.filepath, self.gpu_index, self.plot = pollTime, active, filepath, gpu_index, plot
        self.daemon = True      # exits with the main thread when it crashes or is interrupted

    def run(self):
        import psutil
        proc = psutil.Process()
        timeCounter = 0.0
        with open(self.filepath, "w+", newline="") as f:
            csv.writer(f).writerow(HEADER)
        while self.active:
            time.sleep(self.pollTime)
            timeCounter += self.pollTime
            with open(self.filepath, "a", newline="") as f:
                csv.writer(f).writerow([round(timeCounter, 6), psutil.cpu_percent(), round(proc.memory_info().rss / (1024 ** 2)),
                                        round(gpu_busy_percent(self.gpu_index), 1)])
        if self.plot:
            create_graph_from_csv(self.filepath)


def create_graph_from_csv(filepath):
    """benchmarkLogger.py:41-80: CPU / memory / GPU curves over time next to the CSV (skipped without matplotlib)."""
    try:
        import matplotlib
        matplotlib.use("Agg")
        from matplotlib import pyplot
    except ImportError:
        return None
    with open(filepath) as f:
        rows = list(csv.reader(f))[1:]
    if not rows:
        return None
    t = [float(r[0]) for r in rows]
    fig, axes = pyplot.subplots(3, 1, sharex=True, figsize=(8, 8))
    for ax, col, name in zip(axes, (1, 2, 3), HEADER[1:]):
        ax.plot(t, [float(r[col]) for r in rows])
        ax.set_ylabel(name)
    axes[-1].set_xlabel(HEADER[0])
    out = os.path.splitext(filepath)[0] + ".png"
    fig.savefig(out)
    pyplot.close(fig)
    return out
