"""Regenerates the ctypes stub of INTEGRATION.md section 1 from include/binrec.h (the same parser the package binds with), so the
example cannot drift from the header; tests/test_cabi_cpu.py checks that the committed text is what this prints.
  python tools/gen_integration_stub.py            # print the block
  python tools/gen_integration_stub.py --write    # rewrite it in INTEGRATION.md (between the BEGIN / END markers)"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BEGIN, END = "<!-- BEGIN generated stub (tools/gen_integration_stub.py) -->", "<!-- END generated stub -->"
FN = "brNeumfEmbedForward"


def prototype_text(name):
    src = open(os.path.join(ROOT, "include", "binrec.h")).read()
    m = re.search(r"\bint\s+" + name + r"\s*\(([^;]*?)\)\s*;", src, flags=re.S)
    return "int " + name + "(" + " ".join(m.group(1).split()) + ");"


def stub():
    from importlib import import_module
    _lib = import_module("binary-recommendation_amd._lib")
    import ctypes
    restype, argtypes, argnames = _lib.parse_header()[FN]
    tn = {ctypes.c_void_p: "ctypes.c_void_p", ctypes.c_int64: "ctypes.c_int64", ctypes.c_float: "ctypes.c_float", ctypes.c_double: "ctypes.c_double",
          ctypes.c_uint32: "ctypes.c_uint32", ctypes.c_uint64: "ctypes.c_uint64", ctypes.c_int32: "ctypes.c_int32", ctypes.c_int: "ctypes.c_int"}
    proto = prototype_text(FN)
    wrapped, line = [], "#"
    for w in proto.split(" "):
        if len(line) + 1 + len(w) > 110:
            wrapped.append(line); line = "#    "
        line += " " + w
    wrapped.append(line)
    args = "\n".join(f"    {tn[t]},{' ' * (18 - len(tn[t]))}# {n}" for t, n in zip(argtypes, argnames))
    call = {"user_mlp": "user_mlp.data_ptr()", "item_mlp": "item_mlp.data_ptr()", "user_mf": "user_mf.data_ptr()", "item_mf": "item_mf.data_ptr()",
            "ld_user": "user_mlp.stride(0)", "ld_item": "item_mlp.stride(0)", "user_rows": "user_mlp.shape[0]", "item_rows": "item_mlp.shape[0]",
            "users": "users.data_ptr()", "items": "items.data_ptr()", "id_type": "0", "dim": "user_mlp.shape[1]", "batch": "users.shape[0]",
            "item_first": "1", "x0": "x0.data_ptr()", "dot": "dot.data_ptr()", "err_flag": "None", "stream": "torch.cuda.current_stream().cuda_stream"}
    note = {"id_type": "BR_IDS_I32", "item_first": "NFC_plain.py:137 concat [item, user]", "ld_user": "row stride in floats (2 * dim for fused [mlp | mf] rows)"}
    missing = [n for n in argnames if n not in call]
    assert not missing, f"gen_integration_stub: no example value for {missing}"
    calls = "\n".join(f"        {call[n]},{' ' * max(1, 44 - len(call[n]))}# {n}" + (f": {note[n]}" if n in note else "") for n in argnames)
    return f"""```python
import ctypes, torch
lib = ctypes.CDLL("binary-recommendation_amd/libbinrec_hip.so")

{chr(10).join(wrapped)}
lib.{FN}.restype = ctypes.c_int
lib.{FN}.argtypes = [                     # {len(argtypes)} arguments, in header order
{args}
]

def embed_forward(user_mlp, item_mlp, user_mf, item_mf, users, items, x0, dot):   # torch CUDA tensors (fp32 tables, int32 ids)
    rc = lib.{FN}(
{calls}
    )
    if rc: raise RuntimeError(lib.brGetLastError().decode())
```"""


if __name__ == "__main__":
    text = stub()
    if "--write" in sys.argv:
        p = os.path.join(ROOT, "INTEGRATION.md")
        s = open(p).read()
        i0, i1 = s.index(BEGIN) + len(BEGIN), s.index(END)
        open(p, "w").write(s[:i0] + "\n" + text + "\n" + s[i1:])
    else:
        print(text)
