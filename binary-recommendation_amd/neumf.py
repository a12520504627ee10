"""NeuMF training/inference engine on the HIP hot path (single GPU or one data-parallel rank).

Covers both reference graphs with one parametrisation:
  variant "A": trainers/NFC_plain.py:107-155  (concat [item,user], Dense 100/50/10 sigmoid,
               head concat [mf, mlp], BCE, Adam lr .005)
  variant "B": src/models/NeuMFModel.py:53-100 (concat [user,item], Dense F, F/2, F/4 relu,
               head concat [mlp, mf], MSE, Adam 1e-3)

Every arithmetic op is a libbinrec_hip.so launch (ops.py); torch only owns the buffers, the
stream and (data-parallel) the RCCL collectives.  One training step is the launch sequence in
`NeuMFEngine.train_step` — see DESIGN.md for the kernel-by-kernel data flow.
"""
from __future__ import annotations

import ctypes
import dataclasses
import math
import os

import torch

from . import _lib, ops

DENSE_ORDER = ("W1", "b1", "g1", "be1", "W2", "b2", "g2", "be2", "W3", "b3", "W4", "b4")
TABLES = ("user_mlp", "item_mlp", "user_mf", "item_mf")


@dataclasses.dataclass
class NeuMFConfig:
    variant: str = "A"
    dim: int = 10                       # latent_dim (NFC_plain.py:109) / numFactor (RModel.py:35)
    hidden: tuple | None = None
    lr: float | None = None
    dropout: float = 0.2                # NFC_plain.py:138,141,144 ; NeuMFModel.py:67,71,75
    bn_eps: float = 1e-3                # [TF-sem] Keras BatchNormalization defaults
    bn_momentum: float = 0.99
    beta1: float = 0.9                  # [TF-sem] Keras Adam defaults
    beta2: float = 0.999
    adam_eps: float = 1e-7
    optimizer: str = "adam_dense"       # "adam_dense" = Keras semantics (non-lazy sparse apply);
                                        # "adam_lazy"  = touched rows only (throughput deviation)
    dense_impl: str = "deferred"        # how adam_dense reaches the untouched rows: "deferred" = per-row catch-up
                                        # replay (same values, no per-step table sweep; include/binrec.h
                                        # "Deferred dense Adam"), "sweep" = one pass over the table per step
    replay: str | None = None           # deferred tables: "fast" (default; BR_REPLAY env overrides the default) = the cheaper form of the
                                        # g = 0 recurrence (include/binrec.h BR_REPLAY_FAST: ~1e-6 of a row's movement off the sweep),
                                        # "exact" = the sweep's own fp32 operations (tables bit-equal to dense_impl="sweep")
    seed: int = 0x5EED
    sync_bn: bool = True                # data-parallel: batch statistics over the GLOBAL batch

    def __post_init__(self):
        assert self.variant in ("A", "B")
        assert self.optimizer in ("adam_dense", "adam_lazy")
        assert self.dense_impl in ("deferred", "sweep")
        if self.replay is None:
            self.replay = os.environ.get("BR_REPLAY", "fast")
        assert self.replay in ("fast", "exact")
        if self.variant == "A":
            self.hidden = tuple(self.hidden) if self.hidden else (100, 50, 10)
            self.act, self.item_first, self.mf_first, self.loss = "sigmoid", 1, 1, "bce"
            self.lr = 0.005 if self.lr is None else self.lr
        else:
            self.hidden = tuple(self.hidden) if self.hidden else (self.dim, self.dim // 2, self.dim // 4)
            self.act, self.item_first, self.mf_first, self.loss = "relu", 0, 0, "mse"
            self.lr = 1e-3 if self.lr is None else self.lr

    def dense_shapes(self):
        n1, n2, n3 = self.hidden
        return {"W1": (2 * self.dim, n1), "b1": (n1,), "g1": (n1,), "be1": (n1,),
                "W2": (n1, n2), "b2": (n2,), "g2": (n2,), "be2": (n2,),
                "W3": (n2, n3), "b3": (n3,), "W4": (n3 + 1,), "b4": (1,)}


class _Flat:
    """One flat fp32 buffer with named views (dense parameters / their grads / Adam slots)."""

    def __init__(self, shapes: dict, device):
        self.offsets, off = {}, 0
        for k, shp in shapes.items():
            n = 1
            for s in shp:
                n *= s
            self.offsets[k] = (off, n, shp)
            off += n
        self.numel = off
        self.buf = torch.zeros(off, dtype=torch.float32, device=device)

    def view(self, k):
        off, n, shp = self.offsets[k]
        return self.buf[off:off + n].view(*shp)

    def slice(self, k0, k1):
        """contiguous span covering parameters k0..k1 (adjacent in the layout)."""
        o0 = self.offsets[k0][0]
        o1, n1, _ = self.offsets[k1]
        return self.buf[o0:o1 + n1]


class NeuMFEngine:
    def __init__(self, cfg: NeuMFConfig, num_user_rows: int, num_item_rows: int, device, max_batch: int,
                 id_dtype=torch.int32, init_seed: int = 0, dist=None):
        self.cfg, self.device, self.max_batch, self.id_dtype = cfg, torch.device(device), int(max_batch), id_dtype
        self.num_user_rows, self.num_item_rows = int(num_user_rows), int(num_item_rows)
        self.dist = dist                      # None, or the parallel.DistCtx of the row-sharded subclass
        if dist is not None and not self.sharded:
            # replicated tables would need every rank's row gradients (an all-gather of B x 2*dim per stream): the
            # multi-GPU form of this engine is the row-sharded one
            raise ValueError("a process group is only supported through parallel.make_sharded_engine (row-sharded tables)")
        D, (n1, n2, n3) = cfg.dim, cfg.hidden
        if 2 * D > 256 or max(n1, n2) > 128 or n3 > 32:
            raise ValueError("tower widths: 2*dim <= 256 (first layer, split-K above 128), n1, n2 <= 128 and n3 <= 32 in this build")
        dev = self.device
        g = torch.Generator(device="cpu").manual_seed(init_seed)
        # [TF-sem] Embedding init U(-0.05, 0.05); Dense glorot-uniform; bias 0; BN gamma 1 beta 0
        self.deferred = cfg.optimizer == "adam_dense" and cfg.dense_impl == "deferred"
        self._stale = False                   # deferred: rows lag behind self.t until flush()
        self._flush_t = 0
        self._init_tables(g, init_seed)
        self.theta = _Flat(cfg.dense_shapes(), dev)
        for k, (_, _, shp) in self.theta.offsets.items():
            if k.startswith("W"):
                fi, fo = (shp[0], shp[1]) if len(shp) == 2 else (shp[0], 1)
                lim = math.sqrt(6.0 / (fi + fo))
                self.theta.view(k).copy_(((torch.rand(*shp, generator=g) * 2 - 1) * lim).to(dev))
            elif k.startswith("g"):
                self.theta.view(k).fill_(1.0)
        self.grad = _Flat(cfg.dense_shapes(), dev)
        self.adam_m = _Flat(cfg.dense_shapes(), dev)
        self.adam_v = _Flat(cfg.dense_shapes(), dev)
        self.fused_m = {k: torch.zeros_like(v) for k, v in self.fused.items()}
        self.fused_v = {k: torch.zeros_like(v) for k, v in self.fused.items()}
        self._tab_m = {"user_mlp": self.fused_m["user"][:, :D], "user_mf": self.fused_m["user"][:, D:],
                       "item_mlp": self.fused_m["item"][:, :D], "item_mf": self.fused_m["item"][:, D:]}
        self._tab_v = {"user_mlp": self.fused_v["user"][:, :D], "user_mf": self.fused_v["user"][:, D:],
                       "item_mlp": self.fused_v["item"][:, :D], "item_mf": self.fused_v["item"][:, D:]}
        self.moving_buf = torch.zeros(2 * n1 + 2 * n2, device=dev)      # [mm1 | mv1 | mm2 | mv2]
        self.moving = {"mm1": self.moving_buf[:n1], "mv1": self.moving_buf[n1:2 * n1],
                       "mm2": self.moving_buf[2 * n1:2 * n1 + n2], "mv2": self.moving_buf[2 * n1 + n2:]}
        self.moving["mv1"].fill_(1.0)
        self.moving["mv2"].fill_(1.0)
        self.t = 0
        self._alloc(self.max_batch)

    def local_rows(self, name: str) -> int:
        """rows of `name` held by this process (all of them without row-sharding)."""
        return self.num_user_rows if name.startswith("user") else self.num_item_rows

    def _init_tables(self, g, init_seed):
        """HBM layout: ONE allocation per id stream, rows = [mlp (dim) | mf (dim)] (512 B at dim 64), so
        a lookup touches one contiguous row instead of two; `self.tables` exposes the four reference
        tables (NFC_plain.py:115-126) as column views."""
        D, dev = self.cfg.dim, self.device
        self.fused = {}
        for si, stream in enumerate(("user", "item")):
            rows = self.local_rows(stream + "_mf")
            if rows * 2 * D > (1 << 26):   # big tables: draw on the device (same distribution)
                dg = torch.Generator(device=dev).manual_seed(init_seed + 7919 * (1 + si))
                t = torch.rand(rows, 2 * D, generator=dg, device=dev, dtype=torch.float32).mul_(0.1).sub_(0.05)
            else:
                t = (torch.rand(rows, 2 * D, generator=g) * 0.1 - 0.05).to(dev)
            self.fused[stream] = t
        self._make_views()

    def _make_views(self):
        D = self.cfg.dim
        self._tables = {"user_mlp": self.fused["user"][:, :D], "user_mf": self.fused["user"][:, D:],
                        "item_mlp": self.fused["item"][:, :D], "item_mf": self.fused["item"][:, D:]}

    # the four reference tables / their Adam slots as column views; reading them brings a deferred table up
    # to date first (self.fused* are the raw buffers)
    @property
    def tables(self):
        self.flush()
        return self._tables

    @property
    def tab_m(self):
        self.flush()
        return self._tab_m

    @property
    def tab_v(self):
        self.flush()
        return self._tab_v

    def flush(self):
        """Deferred dense Adam: apply the pending g = 0 steps to every row (brAdamFlush).  No-op otherwise."""
        if not (self.deferred and self._stale):
            return
        cfg, lib = self.cfg, _lib.load()
        for k in ("user", "item"):
            t = self.fused[k]
            _lib.check(lib.brAdamFlush(t.data_ptr(), self.fused_m[k].data_ptr(), self.fused_v[k].data_ptr(), self.last[k].data_ptr(),
                                       t.shape[0], t.shape[1], self.step_state.data_ptr(), cfg.beta1, cfg.beta2, cfg.adam_eps, ops._stream()),
                       "brAdamFlush")
        self._stale, self._flush_t = False, self.t

    # ------------------------------------------------------------------ buffers
    def _alloc(self, B):
        cfg, dev = self.cfg, self.device
        D, (n1, n2, n3) = cfg.dim, cfg.hidden
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        self.x0, self.dot = f(B, 2 * D), f(B)
        pad4 = lambda n: (n + 3) & ~3      # rows of the hidden activations / their gradients are padded to 16 B (brNeumfStep)
        self.a1, self.a2, self.a3 = f(B, pad4(n1))[:, :n1], f(B, pad4(n2))[:, :n2], f(B, n3)
        self.logit, self.prob = f(B), f(B)
        self.da3, self.ddot = f(B, n3), f(B)
        self.gh2, self.gh1, self.dx0 = f(B, pad4(n2))[:, :n2], f(B, pad4(n1))[:, :n1], f(B, 2 * D)
        self.g_user, self.g_item = f(B, 2 * D), f(B, 2 * D)      # fused [mlp | mf] row gradients per stream
        # per-step double scratch: [stats1 | stats2 | bsum1 | bsum2], each [BR_STAT_REPLICAS][2n] (the kernels
        # spread their column-sum atomics over 8 replicas; consumers add them)
        R = ops.STAT_REPLICAS
        self.dstat = torch.zeros(R * (4 * n1 + 4 * n2), dtype=torch.float64, device=dev)
        o = 0
        self.stats1 = self.dstat[o:o + R * 2 * n1]; o += R * 2 * n1
        self.stats2 = self.dstat[o:o + R * 2 * n2]; o += R * 2 * n2
        self.bsum1 = self.dstat[o:o + R * 2 * n1]; o += R * 2 * n1
        self.bsum2 = self.dstat[o:o + R * 2 * n2]
        self.msums = torch.zeros(ops.SUM_SLOTS, ops.METRIC_SUMS, dtype=torch.float64, device=dev)   # [slot][loss, se, ae, correct, bce, tp, fp, fn] (epoch)
        self.bn_buf = f(4 * n1 + 4 * n2)      # [scale1|shift1|mean1|rstd1|scale2|shift2|mean2|rstd2]
        self.bn, o = {}, 0
        for k, n in (("scale1", n1), ("shift1", n1), ("mean1", n1), ("rstd1", n1), ("scale2", n2), ("shift2", n2), ("mean2", n2), ("rstd2", n2)):
            self.bn[k] = self.bn_buf[o:o + n]; o += n
        layers = ((2 * D, n1), (n1, n2), (n2, n3))
        # three slab regions [tail | layer 2 | layer 1] (the step driver lays them out; include/binrec.h)
        self.slabs = f(_lib.load().brNeumfStepSlabFloats(B, D, n1, n2, n3))
        self.dz_ws = f(max(ops.dense_backward_ws_floats(B, k, n) for k, n in layers))
        self.nsh = ops.head_slabs(B)
        self.hslabs = f(self.nsh * (n3 + 2))
        self.keep_bits = torch.empty(sum(ops.dropout_keep_words(B, k) for k in (2 * D, n1, n2)), dtype=torch.int32, device=dev)
        self._keep_widths = (2 * D, n1, n2)
        o1, o2 = ops.dropout_keep_words(B, 2 * D), ops.dropout_keep_words(B, 2 * D) + ops.dropout_keep_words(B, n1)
        self._keep_planes = [self.keep_bits[:o1], self.keep_bits[o1:o2], self.keep_bits[o2:]]
        self._keep_for = None                  # (step, row0) whose masks the planes hold (a training step prefetches the next one's)
        self.err = ops.new_err_flag(dev)
        self._alloc_sparse(B)
        self._build_step_struct()

    # ------------------------------------------------------------------ C step driver
    def _build_step_struct(self):
        """brNeumfStep (include/binrec.h): every pointer the fused step driver needs, filled once."""
        lib = _lib.load()

        class Step(ctypes.Structure):
            _fields_ = _lib.parse_struct("brNeumfStep")
        if ctypes.sizeof(Step) != lib.brNeumfStepSizeof():
            raise _lib.BinrecError("brNeumfStep layout mismatch between include/binrec.h and libbinrec_hip.so")
        cfg = self.cfg
        st = Step()
        st.user_rows, st.item_rows = self.local_rows("user_mf"), self.local_rows("item_mf")
        st.dim, (st.n1, st.n2, st.n3) = cfg.dim, cfg.hidden
        st.act, st.loss = ops.ACT[cfg.act], ops.LOSS[cfg.loss]
        st.item_first, st.mf_first = cfg.item_first, cfg.mf_first
        st.id_type = ops.I64 if self.id_dtype == torch.int64 else ops.I32
        # (row-sharded: the table optimizer runs from Python on the owner's shard, the driver never sees it)
        st.adam_dense = (2 if self.deferred else 1 if cfg.optimizer == "adam_dense" else 0) if not self.sharded else 0
        st.dropout, st.bn_eps, st.bn_momentum = cfg.dropout, cfg.bn_eps, cfg.bn_momentum
        st.seed = cfg.seed
        st.bn_local = 1 if (self.dist is not None and not cfg.sync_bn) else 0
        # a data-parallel host all-reduces `grad` between the reduction and the optimizer: per-replica BatchNorm reduces everything in one
        # launch at BNG (2); with global BatchNorm statistics dgamma / dbeta are already global when BNG writes them, AFTER the all-reduce of
        # the locally reduced dW (0: per-phase reductions)
        st.fused_final = 1 if self.dist is None else (0 if cfg.sync_bn else 2)
        st.beta1, st.beta2, st.adam_eps = cfg.beta1, cfg.beta2, cfg.adam_eps
        P = lambda t: t.data_ptr()
        st.user_tab, st.user_m, st.user_v = P(self.fused["user"]), P(self.fused_m["user"]), P(self.fused_v["user"])
        st.item_tab, st.item_m, st.item_v = P(self.fused["item"]), P(self.fused_m["item"]), P(self.fused_v["item"])
        if st.adam_dense == 1:
            st.user_mark, st.item_mark = P(self.user_mark), P(self.item_mark)
        if self.deferred:
            if not self.sharded:
                st.user_last, st.item_last = P(self.last["user"]), P(self.last["item"])
            self._alloc_step_state(st)
        st.theta, st.grad, st.adam_m, st.adam_v = P(self.theta.buf), P(self.grad.buf), P(self.adam_m.buf), P(self.adam_v.buf)
        st.moving = P(self.moving_buf)
        for k in ("x0", "dot", "a1", "a2", "a3", "logit", "prob", "da3", "ddot", "gh2", "gh1", "dx0", "g_user", "g_item"):
            setattr(st, k, P(getattr(self, k)))
        st.bn, st.dstat, st.msums = P(self.bn_buf), P(self.dstat), P(self.msums)
        st.slabs, st.hslabs, st.err_flag, st.dz_ws = P(self.slabs), P(self.hslabs), P(self.err), P(self.dz_ws)
        st.keep_bits = P(self.keep_bits)
        self._bind_indexes(st)
        if not self.sharded and os.environ.get("BR_AUX_STREAM", "1") != "0":
            self.aux_stream = torch.cuda.Stream(device=self.device)     # the dedup sorts run beside fwd/bwd
            st.aux_stream = self.aux_stream.cuda_stream
        self.step_struct = st
        PH = self.PH = {k[6:]: v for k, v in _lib.parse_enums().items() if k.startswith("BR_PH_")}
        PH["OPT_ROWS"] = PH["ROWS_USER"] | PH["SWEEP_USER"] | PH["ROWS_ITEM"] | PH["SWEEP_ITEM"]
        self._graph = None
        self._graph_multi = None

    def _bind_indexes(self, st):
        ui, ii = self.user_index, self.item_index
        st.u_sorted_ids, st.u_sorted_pos, st.u_ws, st.u_ws_bytes = ui.sorted_ids.data_ptr(), ui.sorted_pos.data_ptr(), ui.ws.data_ptr(), ui.ws_bytes
        st.i_sorted_ids, st.i_sorted_pos, st.i_ws, st.i_ws_bytes = ii.sorted_ids.data_ptr(), ii.sorted_pos.data_ptr(), ii.ws.data_ptr(), ii.ws_bytes
        st.u_seg_ws, st.i_seg_ws = ui.seg_ws(2 * self.cfg.dim).data_ptr(), ii.seg_ws(2 * self.cfg.dim).data_ptr()

    def _run(self, phases: int):
        _lib.check(_lib.load().brNeumfStepRun(ctypes.byref(self.step_struct), phases, ops._stream()), "brNeumfStepRun")

    def _check_batch(self, users, items, labels):
        for t in (users, items):
            if t.dtype != self.id_dtype or not t.is_cuda or not t.is_contiguous():
                raise TypeError(f"ids must be contiguous {self.id_dtype} device tensors")
        if labels is not None and (labels.dtype != torch.float32 or not labels.is_contiguous()):
            raise TypeError("labels must be contiguous float32")

    def _set_batch(self, users, items, labels, B, training, row0, batch_total):
        st = self.step_struct
        self._check_batch(users, items, labels)
        st.batch, st.batch_total, st.row0 = B, batch_total, row0
        st.users, st.items = users.data_ptr(), items.data_ptr()
        st.labels = labels.data_ptr() if labels is not None else None
        st.training, st.step = int(training), self.t
        st.alpha_t = ops.adam_alpha(self.cfg.lr, max(self.t, 1), self.cfg.beta1, self.cfg.beta2)
        # dropout planes: laid out for max_batch rows; a single-process training step prefetches the next step's planes beside its
        # Adam-rows kernel (brNeumfStep.keep_prefetch), and this step skips the generation if it is the one they were made for
        st.keep_rows = self.max_batch
        # 2 (default): the planes and the dense finalize ride in the Adam-rows launch's grid; BR_KEEP_PREFETCH=1: launches of their own
        # on the aux stream beside it (a fork / join inside the step's hipGraph; include/binrec.h keep_prefetch)
        st.keep_prefetch = int(os.environ.get("BR_KEEP_PREFETCH", "2")) if (training and self.dist is None and st.aux_stream) else 0
        st.keep_ready = 1 if (training and self._keep_for == (self.t, row0)) else 0

    def _alloc_sparse(self, B):
        dev = self.device
        self.user_index = ops.RowIndex(B, self.id_dtype, dev)
        self.item_index = ops.RowIndex(B, self.id_dtype, dev)
        if self.deferred:
            self.last = {k: torch.zeros(self.local_rows(k + "_mf"), dtype=torch.int32, device=dev) for k in ("user", "item")}
        elif self.cfg.optimizer == "adam_dense":
            self.user_mark = torch.zeros(self.local_rows("user_mf"), dtype=torch.uint8, device=dev)
            self.item_mark = torch.zeros(self.local_rows("item_mf"), dtype=torch.uint8, device=dev)

    def _alloc_step_state(self, st):
        """device step state (include/binrec.h brStepStateBytes): {step, alpha_t, alpha ring}."""
        if getattr(self, "step_state", None) is None:
            self.step_state = ops.new_step_state(self.device, self.cfg.beta1, self.cfg.beta2, self.cfg.adam_eps, self.cfg.replay)
        st.lr, st.step_state = self.cfg.lr, self.step_state.data_ptr()
        self._sync_step_state()

    ALPHA_RING = _lib.parse_enums()["BR_ALPHA_RING"]
    sharded = False   # the row-sharded subclass runs its own embed exchange (parallel.py)

    def _embed_forward(self, users, items, B):
        raise NotImplementedError   # only the sharded subclass routes the embed block through Python

    # ------------------------------------------------------------------ one optimizer step
    def train_step(self, users, items, labels, row0: int = 0, batch_total: int | None = None):
        """users/items: device int ids (B,), labels: device float32 (B,).  One brNeumfStepRun call on a
        single GPU (no host sync); with a process group the phases are interleaved with the BatchNorm /
        dense-gradient all-reduces and, when row-sharded, the embedding exchange."""
        cfg, PH = self.cfg, self.PH
        B = users.shape[0]
        if B > self.max_batch:
            raise ValueError(f"batch {B} > max_batch {self.max_batch}")
        if B == 0:
            if self.dist is not None and self.dist.world > 1:
                # the peers are about to enter the step's collectives: returning here would leave them waiting forever
                raise ValueError("empty local batch in a data-parallel step: give every rank at least one pair (pad or drop the ragged tail)")
            return
        batch_total = B if batch_total is None else batch_total
        if self.deferred:
            if self.t + 1 - self._flush_t >= self.ALPHA_RING - 8:      # the replay reads alpha_j from a ring
                self.flush()
            self._stale = True
        self.t += 1
        if self._graph is not None and B == self._graph["batch"] and row0 == 0 and batch_total == B:
            self._replay(users, items, labels)
            return
        self._set_batch(users, items, labels, B, True, row0, batch_total)
        if self.dist is None:
            self._run(PH["ALL"])
            self._keep_for = (self.t + 1, row0) if self.step_struct.keep_prefetch and cfg.dropout > 0 else None
            return
        self._dist_step(users, items, B)

    def _dist_step(self, users, items, B):
        """the launches and collectives of one data-parallel step (after _set_batch): no host sync with the fixed-capacity exchange, so the
        row-sharded engine can capture this body into a hipGraph (parallel.py ShardedNeuMFEngine.enable_graph)."""
        cfg, PH = self.cfg, self.PH
        d, sync = self.dist, cfg.sync_bn
        emb = 0 if self.sharded else PH["EMBED"]
        if self.sharded:
            if getattr(self, "step_state", None) is not None:
                # the lookup below replays against the device step counter: advance it first (the driver only does
                # so in a call that holds FWD1|EMBED); the same launch clears the step's double scratch
                _lib.check(_lib.load().brStepStateAdvance(self.step_state.data_ptr(), cfg.lr, cfg.beta1, cfg.beta2, self.dstat.data_ptr(),
                                                          self.dstat.numel(), ops._stream()), "brStepStateAdvance")
            self._embed_forward(users, items, B)
        if not sync:
            # per-replica BatchNorm (what MirroredStrategy does with a plain BatchNormalization [TF-sem]): no collective
            # until the dense gradients, so the whole tower is one driver call
            self._run(PH["FWD1"] | PH["FWD2"] | PH["FWD3"] | PH["BWD2"] | PH["BWD1"] | PH["BNG"] | emb)
            d.all_reduce_sum(self.grad.buf)
        else:
            self._run(PH["FWD1"] | emb)
            d.all_reduce_sum(self.stats1)
            self._run(PH["FWD2"])
            d.all_reduce_sum(self.stats2)
            self._run(PH["FWD3"])
            d.all_reduce_sum(self.bsum2)
            self._run(PH["BWD2"])
            d.all_reduce_sum(self.bsum1)
            self._run(PH["BWD1"])
            # dgamma/dbeta are the BN-backward column sums: global once the sums are, so they are written AFTER
            # the sums' all-reduce and BEFORE nothing else needs them
            d.all_reduce_sum(self.grad.buf)
            self._run(PH["BNG"])
        if self.sharded:
            self._embed_backward_apply(users, items, B)
        else:
            self._run(PH["OPT_TABLES"] | PH["EMBED"] | PH["INDEX"] | PH["OPT_ROWS"])
        self._run(PH["OPT_DENSE"])

    # ------------------------------------------------------------------ hipGraph replay of the step
    PHASE_ORDER = ("FWD1", "FWD2", "FWD3", "BWD2", "BWD1", "BNG", "OPT_TABLES", "ROWS_USER", "SWEEP_USER", "ROWS_ITEM",
                   "SWEEP_ITEM", "OPT_DENSE")

    def enable_graph(self, batch: int | None = None, eager_phases: tuple = (), keep_graph: bool = False):
        """Capture the single-GPU training step for batches of exactly `batch` pairs into a hipGraph and
        replay it from `train_step` (other batch sizes keep the eager launch sequence).  The two per-step
        scalars (dropout step counter, Adam alpha_t) then live in device memory (brNeumfStep.step_state)
        and are advanced by a 1-thread kernel at the top of the step; ids/labels are read from the static
        buffers `in_users / in_items / in_labels` (train_step copies into them unless it is handed
        exactly those tensors).
        eager_phases: consecutive names from PHASE_ORDER that stay OUTSIDE the graphs (graph A -> eager launches
        -> graph B) so that HIP events can bracket them; timed events cannot be recorded inside a capture on
        ROCm 7.2 (bench.py's fallback when the runtime refuses event-record nodes).
        keep_graph: keep the captured hipGraph_t alive beside its executable (torch.cuda.CUDAGraph(keep_graph=True)) - node handles
        placed during the capture (brProbeGraph*) stay valid for hipGraphExec*SetEvent."""
        if self.dist is not None:
            raise ValueError("graph replay covers the single-GPU step (collectives run between the phases otherwise)")
        B = self.max_batch if batch is None else int(batch)
        if not 0 < B <= self.max_batch:
            raise ValueError("graph batch must be in (0, max_batch]")
        dev, st, PH = self.device, self.step_struct, self.PH
        if getattr(self, "in_users", None) is None or self.in_users.shape[0] != B:
            # (kept across captures of the same batch size: an earlier capture that is still replayed reads these very buffers)
            self.in_users = torch.zeros(B, dtype=self.id_dtype, device=dev)
            self.in_items = torch.zeros(B, dtype=self.id_dtype, device=dev)
            self.in_labels = torch.zeros(B, dtype=torch.float32, device=dev)
        self._alloc_step_state(st)
        # every kernel of the step runs once outside a capture first (code objects load on first launch);
        # the model state is put back afterwards
        keep = self._snapshot_for_dry_run()
        keep_sums = self.msums.clone()
        self._sync_step_state()
        self._set_batch(self.in_users, self.in_items, self.in_labels, B, True, 0, B)
        self._run(PH["ALL"])
        torch.cuda.synchronize(dev)
        self._restore_after_dry_run(keep)
        self.msums.copy_(keep_sums)
        del keep
        self._set_batch(self.in_users, self.in_items, self.in_labels, B, True, 0, B)
        st.keep_ready = 1 if st.keep_prefetch else 0      # replays find the planes the previous step (or _replay itself) left
        if eager_phases:
            order = self.PHASE_ORDER
            idx = sorted(order.index(n) for n in eager_phases)
            if idx != list(range(idx[0], idx[-1] + 1)):
                raise ValueError("eager_phases must be consecutive in PHASE_ORDER")
            mask = lambda names: sum(PH[n] for n in names)
            # EMBED modifies FWD1 / OPT_TABLES wherever they land; the dedup sorts go with FWD1 (aux stream)
            parts = [(mask(order[:idx[0]]), True), (mask(order[idx[0]:idx[-1] + 1]), False), (mask(order[idx[-1] + 1:]), True)]
            parts = [(m | PH["EMBED"] | (PH["INDEX"] if m & PH["FWD1"] else 0), cap) for m, cap in parts if m]
        else:
            parts = [(PH["ALL"], True)]
        graphs = []
        for ph, cap in parts:
            if not cap:
                graphs.append(None)
                continue
            g = torch.cuda.CUDAGraph(keep_graph=True) if keep_graph else torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._run(ph)
            if keep_graph:
                g.instantiate()
            graphs.append(g)
        parts = [ph for ph, _ in parts]
        # the capture itself executes nothing, but the dry run above advanced the device step counter
        self._sync_step_state()
        self._keep_for = None                 # the dry run above left another step's planes
        self._graph = {"batch": B, "parts": parts, "graphs": graphs}

    def disable_graph(self):
        self._graph = None
        self._graph_multi = None

    def enable_graph_multi(self, batch: int | None = None, steps: int = 4, keep_graph: bool = False, probe_tag: int | None = None):
        """Capture `steps` consecutive training steps as ONE hipGraph (`train_steps` replays it): a training loop whose batches are already
        on the device - `fit()` over resident arrays, bench.py - then pays the per-launch costs of a graph (the gap between two graph
        launches, ~8 us here, and the staging launch) once per `steps` steps instead of once per step.  Everything that differs between
        the steps of a group lives on the device (step counter, alpha_t, the alpha ring, the dropout planes the previous step's
        Adam-rows launch prefetched); the group's ids / labels are staged by one brStageBatch launch into static buffers of `steps`
        batches.  Needs the single-step graph (captured here if absent: it runs the dry step that loads every code object).
        probe_tag: bench.py's measurement - only the LAST step of the group carries the event-record nodes around that kernel."""
        S = int(steps)
        if S < 2:
            raise ValueError("enable_graph_multi: steps >= 2 (enable_graph is the single-step form)")
        B = self.max_batch if batch is None else int(batch)
        if self._graph is None or self._graph["batch"] != B or len(self._graph["graphs"]) != 1 or self._graph["graphs"][0] is None:
            self.enable_graph(B)
        dev, st, PH = self.device, self.step_struct, self.PH
        gm = getattr(self, "_graph_multi", None)
        if gm is None or gm["S"] != S or gm["batch"] != B:
            self.in_users_m = torch.zeros(S * B, dtype=self.id_dtype, device=dev)
            self.in_items_m = torch.zeros(S * B, dtype=self.id_dtype, device=dev)
            self.in_labels_m = torch.zeros(S * B, dtype=torch.float32, device=dev)
        lib = _lib.load()
        g = torch.cuda.CUDAGraph(keep_graph=True) if keep_graph else torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g):
                for k in range(S):
                    sl = slice(k * B, (k + 1) * B)
                    self._set_batch(self.in_users_m[sl], self.in_items_m[sl], self.in_labels_m[sl], B, True, 0, B)
                    st.keep_ready = 1 if st.keep_prefetch else 0
                    if probe_tag is not None and k == S - 1:
                        lib.brProbeGraphSelect(int(probe_tag))
                    self._run(PH["ALL"])
        finally:
            if probe_tag is not None:
                lib.brProbeGraphSelect(-1)
        if keep_graph:
            g.instantiate()
        self._set_batch(self.in_users, self.in_items, self.in_labels, B, True, 0, B)     # the single-step graph's view of the step struct
        st.keep_ready = 1 if st.keep_prefetch else 0
        self._graph_multi = {"S": S, "batch": B, "graph": g}
        return self._graph_multi

    def train_steps(self, users, items, labels):
        """`steps` training steps in one graph launch (enable_graph_multi): users / items / labels hold the group's batches back to back,
        (steps * batch,) or (steps, batch), contiguous device tensors.  Equal to `steps` calls of train_step on the slices."""
        gm = getattr(self, "_graph_multi", None)
        if gm is None:
            raise RuntimeError("train_steps: enable_graph_multi first")
        S, B = gm["S"], gm["batch"]
        if users.numel() != S * B or items.numel() != S * B or labels.numel() != S * B:
            raise ValueError(f"train_steps: {S} batches of {B} pairs expected")
        self._check_batch(users, items, labels)
        cfg = self.cfg
        if self.deferred:
            if self.t + S - self._flush_t >= self.ALPHA_RING - 8:      # the replay reads alpha_j from a ring
                self.flush()
            self._stale = True
        if users.data_ptr() != self.in_users_m.data_ptr() or items.data_ptr() != self.in_items_m.data_ptr() or labels.data_ptr() != self.in_labels_m.data_ptr():
            _lib.check(_lib.load().brStageBatch(self.in_users_m.data_ptr(), self.in_items_m.data_ptr(), self.in_labels_m.data_ptr(), users.data_ptr(),
                                                items.data_ptr(), labels.data_ptr(), self.step_struct.id_type, S * B, ops._stream()), "brStageBatch")
        prefetching = cfg.dropout > 0 and self.step_struct.aux_stream
        if prefetching and self._keep_for != (self.t + 1, 0):
            ops.dropout_keep_bits(cfg.dropout, cfg.seed, self.t + 1, 0, self.max_batch, (0, 1, 2), self._keep_widths, self._keep_planes)
        self.t += S
        gm["graph"].replay()
        self._keep_for = (self.t + 1, 0) if prefetching else None

    def _snapshot_for_dry_run(self):
        """what enable_graph's dry-run step can change.  Per-step sweep: every table row moves, so the whole (flushed) state.  Deferred /
        lazy tables: only the rows the static input buffers name move - those rows with their moments and `last`, the dense state and
        the host counters are kept, NOT flushed: a capture in the middle of a run leaves every other row's lag as it is (state_dict()
        would flush the tables: 42 GB cloned and every lag reset at config 5's shard size)."""
        if self.cfg.optimizer == "adam_dense" and not self.deferred:
            return {"full": {k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.state_dict().items()}}
        snap = {"rows": {}, "host": (self.t, self._flush_t, self._stale)}
        for k, ids in (("user", self.in_users), ("item", self.in_items)):
            idx = torch.unique(ids).long().clamp_(0, self.fused[k].shape[0] - 1)
            snap["rows"][k] = (idx, self.fused[k][idx], self.fused_m[k][idx], self.fused_v[k][idx], self.last[k][idx] if self.deferred else None)
        snap["dense"] = [t.clone() for t in (self.theta.buf, self.adam_m.buf, self.adam_v.buf, self.moving_buf)]
        return snap

    def _restore_after_dry_run(self, snap):
        if "full" in snap:
            self.load_state_dict(snap["full"])
            return
        for k, (idx, th, m, v, last) in snap["rows"].items():
            self.fused[k][idx] = th; self.fused_m[k][idx] = m; self.fused_v[k][idx] = v
            if last is not None:
                self.last[k][idx] = last
        for dst, src in zip((self.theta.buf, self.adam_m.buf, self.adam_v.buf, self.moving_buf), snap["dense"]):
            dst.copy_(src)
        self.t, self._flush_t, self._stale = snap["host"]
        self._sync_step_state()

    def _sync_step_state(self):
        """device step state := (self.t, alpha_t, beta^t) (after enable_graph / load_state_dict)."""
        if getattr(self, "step_state", None) is not None:
            cfg = self.cfg
            _lib.check(_lib.load().brStepStateSet(self.step_state.data_ptr(), self.t, cfg.lr, cfg.beta1, cfg.beta2, ops._stream()), "brStepStateSet")

    def _replay(self, users, items, labels):
        if users.data_ptr() != self.in_users.data_ptr() or items.data_ptr() != self.in_items.data_ptr() or labels.data_ptr() != self.in_labels.data_ptr():
            self._check_batch(users, items, labels)
            _lib.check(_lib.load().brStageBatch(self.in_users.data_ptr(), self.in_items.data_ptr(), self.in_labels.data_ptr(), users.data_ptr(),
                                                items.data_ptr(), labels.data_ptr(), self.step_struct.id_type, users.shape[0], ops._stream()),
                       "brStageBatch")
        gr = self._graph
        cfg = self.cfg
        prefetching = cfg.dropout > 0 and self.step_struct.aux_stream
        if prefetching and self._keep_for != (self.t, 0):
            # first replay (or the previous step ran another path): the graph expects this step's planes to be there
            ops.dropout_keep_bits(cfg.dropout, cfg.seed, self.t, 0, self.max_batch, (0, 1, 2), self._keep_widths, self._keep_planes)
            self._keep_for = (self.t, 0)
        for ph, g in zip(gr["parts"], gr["graphs"]):
            if g is None:
                self._set_batch(self.in_users, self.in_items, self.in_labels, gr["batch"], True, 0, gr["batch"])
                self.step_struct.keep_ready = 1 if prefetching else 0
                self._run(ph)
            else:
                g.replay()
        self._keep_for = (self.t + 1, 0) if prefetching else None

    def row_grad_views(self, B):
        """name -> (tensor, row stride): the MLP halves are views of dx0, the MF halves of g_user / g_item.
        (Inspection helper for tests.  Single-GPU deferred mode: the step never writes the MF gradients out - the optimizer
        launch forms ddot[b] * partner row as it reads the stashed rows (brAdamRowsSortedPair hi_scale) - so they are
        formed here, with the same single fp32 multiply.)"""
        D = self.cfg.dim
        uo, io = (D, 0) if self.cfg.item_first else (0, D)
        dx0 = self.dx0[:B]
        gu, gi = self.g_user[:B, D:], self.g_item[:B, D:]
        if self.deferred and not self.sharded:
            dd = self.ddot[:B, None]
            gu, gi = dd * gi, dd * gu
        return {"user_mlp": (dx0[:, uo:uo + D], 2 * D), "item_mlp": (dx0[:, io:io + D], 2 * D),
                "user_mf": (gu, 2 * D), "item_mf": (gi, 2 * D)}

    def _embed_backward_apply(self, users, items, B):
        """B1 row gradients of the 4 tables, S1 dedup index, O1 Adam on the (fused) tables."""
        cfg, t, D = self.cfg, self._tables, self.cfg.dim
        ops.neumf_embed_backward(t["user_mf"], t["item_mf"], users, items, cfg.item_first, None, self.ddot[:B],
                                 self.g_user[:B, D:], self.g_item[:B, D:])
        self.user_index.build(users, self.num_user_rows)
        self.item_index.build(items, self.num_item_rows)
        rg = self.row_grad_views(B)
        self._adam_tables({"user": (rg["user_mlp"][0], 2 * D, rg["user_mf"][0], 2 * D),
                           "item": (rg["item_mlp"][0], 2 * D, rg["item_mf"][0], 2 * D)})

    def _adam_tables(self, rg, replayed=None):
        """rg: stream -> (mlp-half grads, stride, mf-half grads, stride), aligned with the positions the
        indexes were built on.  One launch per fused table (+ the dense sweep in Keras mode).
        replayed: stream -> the fused rows as this step's deferred gather wrote them, by the same positions (a row-sharded owner:
        the rows it served) - the optimizer then replays the moments only."""
        cfg, D = self.cfg, self.cfg.dim
        a = ops.adam_alpha(cfg.lr, self.t, cfg.beta1, cfg.beta2)
        hp = dict(beta1=cfg.beta1, beta2=cfg.beta2, eps=cfg.adam_eps)
        dense = cfg.optimizer == "adam_dense" and not self.deferred      # per-step sweep of the untouched rows
        if (self.deferred and replayed is not None and rg["user"][2] is None and rg["item"][2] is None
                and self.user_index.n == self.item_index.n and rg["user"][1] == rg["item"][1] == 2 * D
                and replayed["user"].stride(0) == replayed["item"].stride(0)):
            # fixed-capacity exchange: both streams have the same number of slots -> the two shards in ONE launch (each alone leaves
            # HBM half idle: random 512-B rows behind a dependent chain per row)
            ops.adam_rows_sorted_deferred_pair_replayed(
                self.fused["user"], self.fused_m["user"], self.fused_v["user"], self.last["user"], self.user_index, rg["user"][0], replayed["user"],
                self.fused["item"], self.fused_m["item"], self.fused_v["item"], self.last["item"], self.item_index, rg["item"][0], replayed["item"],
                D, self.step_state, **hp)
            return
        for stream in ("user", "item"):
            idx = self.user_index if stream == "user" else self.item_index
            mark = (self.user_mark if stream == "user" else self.item_mark) if dense else None
            g0, ld0, g1, ld1 = rg[stream]
            if self.deferred:
                ops.adam_rows_sorted_deferred(self.fused[stream], self.fused_m[stream], self.fused_v[stream], self.last[stream], idx, g0, ld0,
                                              self.step_state, row_grads_hi=g1, ldg_hi=ld1, split=D if g1 is not None else 0,
                                              replayed=None if replayed is None else replayed[stream], **hp)
                continue
            ops.adam_rows_sorted(self.fused[stream], self.fused_m[stream], self.fused_v[stream], idx, g0, ld0, a, mark=mark,
                                 row_grads_hi=g1, ldg_hi=ld1, split=D if g1 is not None else 0, **hp)
            if dense:
                ops.adam_dense_sweep(self.fused[stream], self.fused_m[stream], self.fused_v[stream], a, mark=mark, **hp)

    # ------------------------------------------------------------------ inference
    def _infer(self, users, items, labels, n):
        PH = self.PH
        self.flush()
        self._set_batch(users, items, labels, n, False, 0, n)
        if self.sharded:
            self._embed_forward(users, items, n)
            self._run(PH["FWD1"] | PH["FWD2"] | PH["FWD3"])
        else:
            self._run(PH["FWD1"] | PH["FWD2"] | PH["FWD3"] | PH["EMBED"])

    def predict(self, users, items, out=None):
        """sigmoid output of the graph in inference mode (moving BN stats, no dropout)."""
        B = users.shape[0]
        if out is None:
            out = torch.empty(B, dtype=torch.float32, device=self.device)
        for s in range(0, B, self.max_batch):
            e = min(B, s + self.max_batch)
            self._infer(users[s:e], items[s:e], None, e - s)
            out[s:e].copy_(self.prob[:e - s])
        return out

    def evaluate_batch(self, users, items, labels):
        """inference-mode forward + loss/metric sums accumulated into self.msums (no grads)."""
        self._infer(users, items, labels, users.shape[0])

    def pop_metrics(self, n_samples: int) -> dict:
        """Host sync: the compiled metrics since the last call - mean loss / mse / mae / binary_accuracy (RModel.py:20) and the
        rest of trainers/NFC_plain.py:155: binary_crossentropy, true/false positives/negatives at threshold 0.5."""
        s = self.msums.sum(dim=0).cpu().tolist()
        self.msums.zero_()
        n = max(1, n_samples)
        return {"loss": s[0] / n, "mse": s[1] / n, "mae": s[2] / n, "binary_accuracy": s[3] / n, "binary_crossentropy": s[4] / n,
                "true_positives": s[5], "false_positives": s[6], "false_negatives": s[7], "true_negatives": n_samples - s[5] - s[6] - s[7]}

    def check_ids(self):
        ops.raise_if_flag(self.err)

    # ------------------------------------------------------------------ state
    def state_dict(self) -> dict:
        self.flush()
        sd = {"t": self.t, "theta": self.theta.buf, "adam_m": self.adam_m.buf, "adam_v": self.adam_v.buf}
        for k in ("user", "item"):
            sd["table." + k], sd["table." + k + ".m"], sd["table." + k + ".v"] = self.fused[k], self.fused_m[k], self.fused_v[k]
        sd.update(self.moving)
        return sd

    def load_state_dict(self, sd: dict):
        self.t = int(sd["t"])
        for k, dst in (("theta", self.theta.buf), ("adam_m", self.adam_m.buf), ("adam_v", self.adam_v.buf)):
            dst.copy_(sd[k])
        for k in ("user", "item"):
            self.fused[k].copy_(sd["table." + k]); self.fused_m[k].copy_(sd["table." + k + ".m"]); self.fused_v[k].copy_(sd["table." + k + ".v"])
        for k in self.moving:
            self.moving[k].copy_(sd[k])
        if self.deferred:                   # a checkpoint holds flushed tables: every row includes step t
            for k in ("user", "item"):
                self.last[k].fill_(self.t)
            self._stale, self._flush_t = False, self.t
        self._sync_step_state()

    def load_numpy_params(self, p: dict):
        """Load a parameter dict in the oracle's naming (tests / golden fixtures)."""
        for k in TABLES:
            self.tables[k].copy_(torch.as_tensor(p[k], dtype=torch.float32))
        for k in DENSE_ORDER:
            self.theta.view(k).copy_(torch.as_tensor(p[k], dtype=torch.float32).reshape(self.theta.view(k).shape))
        for k in self.moving:
            self.moving[k].copy_(torch.as_tensor(p[k], dtype=torch.float32))
