"""Print the kernel timeline of the last whole-step graph replays from a rocprofv3 kernel_trace.csv
(start offset, duration, gap to the previous kernel end on the device clock)."""
import csv, sys, glob, re

path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
names = [r[2] for r in rows]
# a step starts at its staging launch (stage_batch_kernel; builds whose chain still has a step_state_advance_kernel of its own: there)
starts = [i for i, n in enumerate(names) if "stage_batch_kernel" in n]
if len(starts) < 40:
    starts = [i for i, n in enumerate(names) if "step_state_advance" in n]
want = int(sys.argv[2]) if len(sys.argv) > 2 else -30     # which step (index into starts)
i0, i1 = starts[want], starts[want + 1]
t0 = rows[i0][0]
prev_end = t0
short = lambda n: re.sub(r"\(.*", "", re.sub(r"void |br::|hipcub::|rocprim::|detail::", "", n))[:70]
busy = 0
for s, e, n in rows[i0:i1]:
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:7.1f}  {short(n)}")
    prev_end = max(prev_end, e)
    busy += e - s
print(f"step span {(prev_end - t0) / 1e3:.1f} us, sum of kernel durations {busy / 1e3:.1f} us, next step starts at {(rows[i1][0] - t0) / 1e3:.1f} us")
