"""-m "not gpu": the C-ABI shared library loads without a GPU and exports every symbol that
include/binrec.h declares (no compute calls here)."""
import ctypes
import os
import re
from importlib import import_module

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    b = import_module("binary-recommendation_amd.build")
    return b.build_library(verbose=False)


def test_header_parses_all_entry_points():
    lib = import_module("binary-recommendation_amd._lib")
    protos = lib.parse_header()
    must = {"brGatherRows", "brRowDot", "brRowDotBackward", "brNeumfEmbedForward", "brNeumfEmbedBackward", "brBprForwardBackward",
            "brRowIndexWorkspaceBytes", "brRowIndexBuild", "brSegmentSumRows", "brScatterAddRows", "brAdamRowsSorted",
            "brAdamDenseSweep", "brAdamFlat", "brAdagradRowsSorted", "brAdagradFlat", "brDenseForward", "brDenseBackward",
            "brDenseBackwardSlabs", "brReduceSlabs", "brBnFinalize", "brBnInference", "brBnParamGrads", "brNeumfHead", "brHeadSlabs",
            "brBceLogits", "brInBatchSoftmaxLse", "brInBatchSoftmaxGrad", "brTopKRows", "brGetLastError", "brVersion", "brDeviceInfo",
            # round 2
            "brDropoutKeepBits", "brDropoutKeepWords", "brNeumfTailFused", "brInBatchSoftmaxLseGradQ", "brInBatchSoftmaxWorkspaceBytes", "brBootstrapDataset",
            "brBprSampleTriplets", "brNcfNegativeCandidates", "brFullAuc", "brMapAtK", "brShardPlanPair", "brShardPadPair", "brRowsToSlotsPair",
            "brAdamRowsSortedPair", "brAdamRowsSortedDeferred", "brAdamFlush", "brGatherRowsDeferred", "brDenseFinalize", "brNeumfStepRun",
            "brAdamRowsSortedDeferredReplayed", "brAdamRowsSortedPairReplayed", "brGatherRowsDeferredPair", "brProbeGraphSelect", "brProbeGraphNodes", "brProbeGraphEnable", "brProbeGraphArm", "brProbeGraphRead"}
    assert must <= set(protos), must - set(protos)
    # pointer / scalar classification sanity
    rt, args, names = protos["brGatherRows"]
    assert rt is ctypes.c_int and len(args) == 10 and args[0] is ctypes.c_int and args[1] is ctypes.c_void_p
    assert protos["brRowIndexWorkspaceBytes"][0] is ctypes.c_int64
    assert protos["brGetLastError"][0] is ctypes.c_char_p


def test_library_exports_every_declared_symbol(built):
    lib = import_module("binary-recommendation_amd._lib")
    cdll = ctypes.CDLL(built)
    for name in lib.parse_header():
        assert hasattr(cdll, name), f"{name} declared in include/binrec.h but not exported"
    handle = lib.load()
    assert handle.brVersion() >= 100
    assert handle.brHeadSlabs(65536) == 256 and handle.brDenseBackwardSlabs(65536, 128, 100) >= 1   # host-only helpers


def test_argument_errors_are_reported_not_crashed(built):
    lib = import_module("binary-recommendation_amd._lib")
    h = lib.load()
    rc = h.brRowDot(None, None, None, 64, 10, None)          # null pointers -> BR_ERR_ARG before any launch
    assert rc == -1 and b"brRowDot" in h.brGetLastError()
    rc = h.brDenseForward(None, 0, None, None, None, 0, 0, 1, 1, 0, None, None, None, 0.0, None, None, None)
    assert rc == -1
    with pytest.raises(lib.BinrecError):
        lib.check(rc, "brDenseForward")
    # entry points added in round 2: argument checks run before anything touches a device
    assert h.brInBatchSoftmaxLseGradQ(None, None, None, None, 0, 8, 8, 64, 0, None, None, None, None, 0, None) == -1 and b"LseGradQ" in h.brGetLastError()
    assert h.brInBatchSoftmaxLse(None, None, None, None, 0, 8, 8, 200, 0, None, None, None, 0, None) == -1         # dim > 128
    assert h.brShardPadPair(None, None, None, None, None, None, None, None, 0, 8, 2, 0, 10, 10, None, None, None, None, None, None, None, None, 0, None, None) == -1
    assert h.brRowsToSlotsPair(None, None, 6, None, None, None, None, 4, 6, None) == -1                               # dim % 4 != 0
    assert h.brFullAuc(None, 0, None, None, 1, 1, None, None) == -1 and h.brMapAtK(None, 1, 0, None, None, None, None, None) == -1
    assert h.brDropoutKeepBits(1.5, 0, 0, 0, 4, 1, None, None, None, None) == -1                                      # drop_p out of range
    assert h.brInBatchSoftmaxWorkspaceBytes(8192, 8192, 64) > 0 and h.brInBatchSoftmaxWorkspaceBytes(0, 8, 64) == 0
    assert h.brDropoutKeepWords(65536, 100) == 65536 * 4
    # replayed-rows optimizer entry and the graph-resident probe
    assert h.brAdamRowsSortedDeferredReplayed(None, None, None, None, 10, 64, None, 0, None, 4, None, 64, None, 0, 0, None, 64, None, 0.9, 0.999, 1e-7, None, None) == -1
    assert b"Replayed" in h.brGetLastError()
    assert h.brProbeGraphNodes() == 0 and h.brProbeGraphArm(None, 0) == -1 and h.brProbeGraphRead(0, None) == -1


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    lib = import_module("binary-recommendation_amd._lib")
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(lib.BinrecError):
        lib.load()


def test_product_never_imports_the_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle", flags=re.M)
    pkg = os.path.join(ROOT, "binary-recommendation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                assert not pat.search(open(os.path.join(dirpath, f)).read()), f
    # bench.py may touch the oracle only inside its cpu_baseline leg
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def cpu_baseline"):src.index("def log(")]
    assert len(pat.findall(src)) == len(pat.findall(body)) == 2


def test_integration_stub_matches_the_header():
    """INTEGRATION.md section 1 is generated from include/binrec.h (tools/gen_integration_stub.py): the committed block equals what the
    generator prints now, and its argtypes list has exactly the header's argument count (round 2 shipped a 16-argument example for an
    18-argument prototype)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_stub", os.path.join(ROOT, "tools", "gen_integration_stub.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = text[text.index(gen.BEGIN) + len(gen.BEGIN):text.index(gen.END)].strip()
    assert block == gen.stub().strip(), "INTEGRATION.md stub is stale: run `python tools/gen_integration_stub.py --write`"
    _lib = import_module("binary-recommendation_amd._lib")
    n_header = len(_lib.parse_header()[gen.FN][1])
    assert block.count("ctypes.c_") - 1 == n_header                      # (one more: the restype line)
    assert block.count("# ") >= 2 * n_header
