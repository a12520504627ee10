"""The reference's model surface (src/models/*, trainers/{NFC_plain,twoTower}.py) re-exposed on top of
the HIP engines: same names, argument meaning and return conventions, so a trainer script switches by
changing its import.  Only the hot path is re-implemented; REST, SMB `DataStore`, model plots, Neptune
and SavedModel export are out of scope.

  KerasLikeNeuMF   <- the Keras `Model` built at NFC_plain.py:107-155 / NeuMFModel.py:53-100
                      fit / predict / evaluate / save / load (NFC_plain.py:165-183, RModel.py:130-147)
  RModel, NeuMFModel, BPRModel <- src/models/RModel.py, NeuMFModel.py, BPRModel.py
  TwoTowerModel    <- trainers/twoTower.py:19-111 (computeEmb / computeLoss / train_step / test_step /
                      setCandidates / call)
Broken call sites of the reference (`strategy` NameError, BPR arity, ...) are not reproduced.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import data as _data
from . import ops
from .bpr import BPREngine
from .neumf import NeuMFConfig, NeuMFEngine
from .two_tower import StringLookup, TwoTowerEngine


class History:
    """keras.callbacks.History stand-in: `.history[name]` = per-epoch list (RModel.py:103-106)."""

    def __init__(self):
        self.history = {}

    def add(self, logs: dict):
        for k, v in logs.items():
            self.history.setdefault(k, []).append(v)


def _dist_ctx(distributedConfig=None):
    """The multi-worker switch of RModel.train (src/models/RModel.py:115-121: `if distributedConfig is not None: strategy =
    MultiWorkerMirroredStrategy()`).  Here the workers are the ranks of a torch.distributed process group (one process per GPU, backend
    "nccl" = RCCL, launched with torch.distributed.run): when one is initialised with more than one rank the models build their
    row-sharded engines on it (parallel.py) -> DistCtx, else None.  A distributedConfig without a process group is an error: the
    TF_CONFIG cluster description of the reference has no meaning here (INTEGRATION.md "Multi-worker")."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        from .parallel import DistCtx
        return DistCtx()
    if distributedConfig is not None and not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("distributedConfig given but torch.distributed is not initialised: start one process per GPU with "
                           "`python -m torch.distributed.run --nproc-per-node N ...` and call torch.distributed.init_process_group('nccl') first")
    return None


def _rank_device(device, ctx):
    """one process per GPU: the default device of a rank is cuda:LOCAL_RANK"""
    if ctx is not None and str(device) == "cuda:0" and "LOCAL_RANK" in os.environ and torch.cuda.device_count() > int(os.environ["LOCAL_RANK"]):
        return f"cuda:{int(os.environ['LOCAL_RANK'])}"
    return device


def _rank_slices(n, gb, ctx):
    """[(start, stop, row0, batch_total)] of THIS rank for the global batches of size gb over n samples: every global batch is cut into
    world near-equal consecutive slices (the reference's MirroredStrategy shards each global batch over its replicas [TF-sem]); a ragged
    tail with fewer samples than ranks is dropped (a rank without a pair cannot enter the step's collectives)."""
    out = []
    W, r = (ctx.world, ctx.rank) if ctx is not None else (1, 0)
    for s in range(0, n, gb):
        m = min(gb, n - s)
        if m < W:
            break
        base, extra = divmod(m, W)
        lo = r * base + min(r, extra)
        cnt = base + (1 if r < extra else 0)
        out.append((s + lo, s + lo + cnt, lo, m))
    return out


def _to_dev(a, device, dtype):
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=dtype).contiguous()
    return torch.as_tensor(np.ascontiguousarray(np.asarray(a)), device=device).to(dtype).contiguous()


# the compiled metric lists of the reference, by the names Keras gives them in `history` / `evaluate`
RMODEL_METRICS = ["mse", "mae", "binary_accuracy"]                                   # RModel.METRICS (RModel.py:20)
NFC_PLAIN_METRICS = ["binary_crossentropy", "mse", "mae", "false_negatives", "false_positives", "true_negatives", "true_positives",
                     "binary_accuracy", "top_k_categorical_accuracy"]                   # trainers/NFC_plain.py:155


class KerasLikeNeuMF:
    METRICS = RMODEL_METRICS

    def __init__(self, engine: NeuMFEngine, user_first_inputs: bool = True, metrics=None):
        self.engine = engine
        self.user_first_inputs = user_first_inputs   # Model([customer_input, material_input]) (NFC_plain.py:154)
        self.metrics_names = ["loss"] + list(metrics if metrics is not None else self.METRICS)

    def _metric_values(self, mt: dict) -> list:
        """[loss, *compiled metrics] in compiled order.  top_k_categorical_accuracy (NFC_plain.py:155, k = 10) on a 1-unit
        output is in_top_k(argmax(y_true) = 0, y_pred, 10) over ONE column: 1.0 for every sample [TF-sem] - reported as that
        constant, it carries no information."""
        return [1.0 if k == "top_k_categorical_accuracy" else float(mt[k]) for k in self.metrics_names]

    def _xy(self, x, y=None):
        e = self.engine
        if isinstance(x, dict):
            u = x.get("user", x.get("customer-input")); i = x.get("item", x.get("material-input"))
        else:
            u, i = (x[0], x[1]) if self.user_first_inputs else (x[1], x[0])
        u, i = _to_dev(u, e.device, e.id_dtype), _to_dev(i, e.device, e.id_dtype)
        yy = None if y is None else _to_dev(y, e.device, torch.float32)
        return u.view(-1), i.view(-1), None if yy is None else yy.view(-1)

    def fit(self, x, y=None, epochs=1, batch_size=None, shuffle=True, verbose=0, callbacks=None, validation_data=None,
            seed=0, orders=None):
        """model.fit([users, items], labels, epochs, batch_size, shuffle) (NFC_plain.py:165).  Per epoch the
        sample order is re-drawn (Keras shuffle=True); metrics are read back once per epoch."""
        e = self.engine
        u, i, yy = self._xy(x, y)
        n = u.shape[0]
        bs = int(batch_size or 32)
        if bs > e.max_batch:
            raise ValueError(f"batch_size {bs} > engine max_batch {e.max_batch}")
        # multi-worker (row-sharded engine): batch_size is the PER-REPLICA batch, every rank holds the same samples and takes its slice
        # of each global batch of batch_size x world (global-row dropout masks: row0; loss mean over the global batch: batch_total)
        ctx = e.dist if getattr(e, "sharded", False) else None
        slices = _rank_slices(n, bs * (ctx.world if ctx is not None else 1), ctx)
        hist = History()
        g = torch.Generator(device=e.device).manual_seed(seed)
        for ep in range(epochs):
            if orders is not None:       # caller-supplied sample order per epoch (reproducible comparisons)
                perm = _to_dev(orders[ep], e.device, torch.int64)
                uu, ii, ll = u[perm], i[perm], yy[perm]
            elif shuffle:
                perm = torch.randperm(n, device=e.device, generator=g)
                uu, ii, ll = u[perm], i[perm], yy[perm]
            else:
                uu, ii, ll = u, i, yy
            gm = getattr(e, "_graph_multi", None) if ctx is None else None
            k = 0
            while k < len(slices):
                lo, hi, row0, bt = slices[k]
                # engine.enable_graph_multi(batch_size, S): S consecutive full batches of the (contiguous) epoch order are one graph launch
                if gm is not None and gm["batch"] == bs and k + gm["S"] <= len(slices) and slices[k + gm["S"] - 1][1] - lo == gm["S"] * bs:
                    end = lo + gm["S"] * bs
                    e.train_steps(uu[lo:end], ii[lo:end], ll[lo:end])
                    k += gm["S"]
                    continue
                e.train_step(uu[lo:hi], ii[lo:hi], ll[lo:hi], row0=row0, batch_total=bt)
                k += 1
            e.check_ids()
            if ctx is not None:
                ctx.all_reduce_sum(e.msums)          # the epoch's metric sums of all replicas (a collective: every rank runs fit)
            logs = dict(zip(self.metrics_names, self._metric_values(e.pop_metrics(sum(s[3] for s in slices)))))
            if validation_data is not None:
                vl = self.evaluate(validation_data[0], validation_data[1], batch_size=bs)
                logs.update({"val_" + k: v for k, v in zip(self.metrics_names, vl)})
            hist.add(logs)
            for cb in callbacks or []:
                cb.on_epoch_end(ep, logs)
            if verbose:
                print(f"Epoch {ep + 1}/{epochs} - " + " - ".join(f"{k}: {v:.6f}" for k, v in logs.items()))
        return hist

    def predict(self, x, batch_size=None, verbose=0):
        u, i, _ = self._xy(x)
        return self.engine.predict(u, i).cpu().numpy().reshape(-1, 1)

    def evaluate(self, x, y, batch_size=None, verbose=0, callbacks=None, steps=None):
        """-> [loss, *compiled metrics] (model.evaluate, NFC_plain.py:183; RModel.py:147): `metrics_names` says which -
        RModel.METRICS by default, NFC_PLAIN_METRICS for the graph of trainers/NFC_plain.py:155."""
        e = self.engine
        u, i, yy = self._xy(x, y)
        bs = min(int(batch_size or e.max_batch), e.max_batch)
        e.msums.zero_()
        n = u.shape[0] if steps is None else min(u.shape[0], steps * bs)
        loss_acc = 0.0
        for s in range(0, n, bs):
            m = min(n, s + bs) - s
            e.evaluate_batch(u[s:s + m], i[s:s + m], yy[s:s + m])
        return self._metric_values(e.pop_metrics(n))

    def save(self, path):
        """model.save(path) (NFC_plain.py:166, RModel.py:139): tables, dense params, BN moving stats and
        optimizer slots as one safetensors-style torch file (not a TF SavedModel)."""
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        if getattr(self.engine, "sharded", False):   # every rank owns different table rows: every rank writes its shard (parallel.save_sharded)
            self.engine.save_sharded(path)
            return
        sd = {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in self.engine.state_dict().items()}
        torch.save(sd, path)

    def load_weights(self, path):
        if getattr(self.engine, "sharded", False):
            self.engine.load_sharded(path)
            return
        self.engine.load_state_dict(torch.load(path, map_location=self.engine.device, weights_only=True))

    def summary(self):
        c = self.engine.cfg
        print(f"NeuMF-{c.variant}: dim {c.dim}, tower {2 * c.dim}->{'->'.join(map(str, c.hidden))}->1, loss {c.loss}, {c.optimizer}")


class RModel:
    """src/models/RModel.py: hyper-parameters, paths and the train() orchestration."""
    CUSTOMER_ID = "CUSTOMER_ID"
    PRODUCT_ID = "PRODUCT_ID"
    METRICS = ["mse", "mae", "binary_accuracy"]

    def __init__(self, moduleName, device="cuda:0"):
        self.modelName = moduleName
        self.checkpointPath = "checkpoints/{}/cp".format(self.modelName)
        self.numFactor, self._epochs, self._batchSize, self._validationSteps, self._testSize = 32, 10, 1024, 20, 0.2
        self._model = None
        self.device = device

    epochs = property(lambda s: s._epochs, lambda s, v: setattr(s, "_epochs", v))
    batchSize = property(lambda s: s._batchSize, lambda s, v: setattr(s, "_batchSize", v))
    testSize = property(lambda s: s._testSize, lambda s, v: setattr(s, "_testSize", v))
    validationSteps = property(lambda s: s._validationSteps, lambda s, v: setattr(s, "_validationSteps", v))
    model = property(lambda s: s._model, lambda s, v: setattr(s, "_model", v))

    def compileModel(self, distributedConfig, numUser, numItem, numFactor):
        print("placeholder")
        return None

    def readyToTrain(self):
        return True

    def isMaster(self, taskType, taskId) -> bool:
        return taskType is None or taskType == "chief" or (taskType == "worker" and taskId == 0)

    def getNumberOfWorkers(self, distributedConfig) -> int:
        return len(distributedConfig["cluster"]["worker"])

    def restoreFromLatestCheckPoint(self):
        self.model.load_weights(self.checkpointPath)


class NeuMFModel(RModel):
    """src/models/NeuMFModel.py."""

    def __init__(self, device="cuda:0", max_batch=65536, optimizer="adam_dense"):
        super().__init__("NeuMFModel", device)
        self.max_batch, self.optimizer = max_batch, optimizer

    def readData(self, path, rowLimit):
        import pandas as pd
        df = pd.read_csv(path, nrows=rowLimit)
        df = df.drop(columns=[c for c in ("MATERIAL", "QUANTITY") if c in df.columns])
        return int(df.PRODUCT_ID.max()) + 1, int(df.CUSTOMER_ID.max()) + 1, df

    def compileModel(self, distributedConfig, numUser: int, numItem: int, numFactor: int):
        """NeuMFModel.py:53-100: relu tower F -> F/2 -> F/4, Dot GMF, MSE, Adam(1e-3)."""
        cfg = NeuMFConfig(variant="B", dim=numFactor, optimizer=self.optimizer)
        ctx = _dist_ctx(distributedConfig)
        if ctx is not None:      # RModel.py:115-121: multi-worker -> tables row-sharded over the ranks, dense parameters data-parallel
            from .parallel import make_sharded_engine
            self.device = _rank_device(self.device, ctx)
            eng = make_sharded_engine(NeuMFEngine)(cfg, numUser, numItem, self.device, self.max_batch, ctx, id_dtype=torch.int32)
        else:
            eng = NeuMFEngine(cfg, numUser, numItem, self.device, self.max_batch, id_dtype=torch.int32)
        self.model = KerasLikeNeuMF(eng)
        return self.model

    def bootstrapDataset(self, df, negRatio=3.0, batchSize=128, shuffle=True, seed=0):
        """NeuMFModel.py:102-123 -> dict(user, item, label, batch_size) of DEVICE tensors instead of a tf.data.Dataset: the
        negatives are sampled on the GPU (csrc/sampling.hip, brBootstrapDataset)."""
        u, i, y = _data.bootstrap_dataset(df[self.CUSTOMER_ID].to_numpy(), df[self.PRODUCT_ID].to_numpy(), negRatio, seed, self.device)
        return {"user": u, "item": i, "label": y, "batch_size": batchSize, "shuffle": shuffle}

    def prepareToTrain(self, distributedConfig, path, rowLimit):
        from sklearn.model_selection import train_test_split
        numItem, numUser, df = self.readData(path, rowLimit)
        trainSplit, testSplit = train_test_split(df, test_size=self.testSize)
        self._products = testSplit.PRODUCT_ID.unique().tolist()
        self._users = testSplit.CUSTOMER_ID.unique().tolist()
        bs = 128 if distributedConfig is None else self.batchSize     # NeuMFModel.py:43-48
        trainDs, testDs = self.bootstrapDataset(trainSplit, batchSize=bs), self.bootstrapDataset(testSplit, batchSize=bs, shuffle=False)
        self.compileModel(distributedConfig, numUser, numItem, self.numFactor)
        return trainDs, testDs, trainSplit

    def train(self, path, rowLimit, metricDict: dict = {}, distributedConfig=None):
        """RModel.train (RModel.py:115-150) -> {'result': 'completed', 'metrics': [...]}."""
        from sklearn.model_selection import train_test_split
        trainDs, testDs, trainSplit = self.prepareToTrain(distributedConfig, path, rowLimit)
        self.model.fit({"user": trainDs["user"], "item": trainDs["item"]}, trainDs["label"], epochs=self.epochs,
                       batch_size=trainDs["batch_size"], shuffle=trainDs["shuffle"],
                       validation_data=({"user": testDs["user"], "item": testDs["item"]}, testDs["label"]))
        os.makedirs(os.path.dirname(self.checkpointPath), exist_ok=True)
        self.model.save(self.checkpointPath)
        _, val = train_test_split(trainSplit, test_size=0.2)
        v = self.bootstrapDataset(val, shuffle=False)
        metrics = self.model.evaluate({"user": v["user"], "item": v["item"]}, v["label"], batch_size=v["batch_size"], steps=self.validationSteps)
        return {"result": "completed", "metrics": metrics}

    def getPredictableUsers(self) -> list:
        return list(self._users)

    def predictForUser(self, customerId, numberOfItem=5, sort="float"):
        """NeuMFModel.py:133-150 -> [(item, score)] as strings, best first.
        The reference sorts the STRING scores (`sorted(extractFeatures.items(), key=lambda x: x[1], reverse=True)`, :150).
        For sigmoid outputs printed in positional notation ("0.73...") that is the numeric order; it differs only where str()
        switches to scientific notation (scores < 1e-4: "9.5e-05" sorts above "0.9").  sort="float" (default) ranks by value -
        what the endpoint means; sort="str" reproduces the reference's lexicographic order exactly."""
        items = np.asarray(self._products)
        p = self.model.predict({"user": np.full(len(items), customerId), "item": items}).reshape(-1)
        if sort == "str":
            feats = {str(items[j]): str(p[j]) for j in range(len(items))}
            return sorted(feats.items(), key=lambda x: x[1], reverse=True)[:numberOfItem]
        order = np.argsort(-p, kind="stable")[:numberOfItem]
        return [(str(items[j]), str(p[j])) for j in order]


class BPRModel(RModel):
    """src/models/BPRModel.py: shared item embedding, triplet loss 1 - sigmoid(u.p - u.n), Adam(1e-3), batch 64."""

    def __init__(self, device="cuda:0", max_batch=65536, optimizer="adam_dense"):
        super().__init__("BPRModel", device)
        self.max_batch, self.optimizer = max_batch, optimizer

    def compileModel(self, distributedConfig, numUser: int, numItem: int, numFactor: int):
        ctx = _dist_ctx(distributedConfig)
        if ctx is not None:
            from .parallel import make_sharded_bpr
            self.device = _rank_device(self.device, ctx)
            self.model = make_sharded_bpr(BPREngine)(numUser, numItem, numFactor, self.device, self.max_batch, ctx, lr=1e-3, optimizer=self.optimizer)
        else:
            self.model = BPREngine(numUser, numItem, numFactor, self.device, self.max_batch, lr=1e-3, optimizer=self.optimizer)
        return self.model, ctx         # the reference returns (model, strategy) (BPRModel.py:74): the process group stands for the strategy

    def extractPositivesNegatives(self, trainDf, customerId, productIds):
        """BPRModel.py:111-119: every (positive, non-interacted) pair of one customer."""
        existing = trainDf[trainDf.CUSTOMER_ID == customerId].PRODUCT_ID.tolist()
        ex = set(existing)
        return [{"CUSTOMER_ID": customerId, "pPRODUCT_ID": p, "nPRODUCT_ID": n} for p in existing for n in productIds if n not in ex]

    def readData(self, path, rowLimit):
        """(numItem, numUser, transactions) like NeuMFModel.readData (the reference's BPRModel has none and fails at
        BPRModel.py:79; RModel.readData returns None)."""
        import pandas as pd
        df = pd.read_csv(path, nrows=rowLimit)
        return int(df.PRODUCT_ID.max()) + 1, int(df.CUSTOMER_ID.max()) + 1, df

    def train(self, path, rowLimit, metricDict: dict = None, distributedConfig=None, negPerPos: int = 4, exhaustive: bool = False, seed: int = 0):
        """BPRModel.train (src/models/BPRModel.py:76-109): batchSize 64, train/test split, triplets, compileModel, fit for
        `epochs`.  The reference enumerates EVERY (positive, non-interacted product) pair of every customer with Pool(5)
        (O(U*I) rows, :94-98,111-119); here `negPerPos` negatives per positive are sampled on the GPU (brBprSampleTriplets);
        exhaustive=True reproduces the enumeration for small data.  -> {'result': 'completed', 'metrics': [last epoch loss]}
        in RModel.train's convention (the reference's BPR train returns None)."""
        from sklearn.model_selection import train_test_split
        self.batchSize = 64
        numItem, numUser, df = self.readData(path, rowLimit)
        self.trainDf, self.testDf = train_test_split(df, test_size=self.testSize, random_state=seed)
        customerIds = self.trainDf.CUSTOMER_ID.unique().tolist()
        self.productIds = self.trainDf.PRODUCT_ID.unique().tolist()
        self.compileModel(distributedConfig, max(customerIds) + 1, max(self.productIds) + 1, self.numFactor)
        tu, ti = self.trainDf.CUSTOMER_ID.to_numpy(), self.trainDf.PRODUCT_ID.to_numpy()
        if exhaustive:
            rows = [r for c in customerIds for r in self.extractPositivesNegatives(self.trainDf, c, self.productIds)]
            X = {"customerId_input": np.array([r["CUSTOMER_ID"] for r in rows], np.float32),
                 "pProduct_input": np.array([r["pPRODUCT_ID"] for r in rows], np.float32),
                 "nProduct_input": np.array([r["nPRODUCT_ID"] for r in rows], np.float32)}
        else:
            u, p, n = _data.sample_bpr_triplets(tu, ti, max(customerIds) + 1, max(self.productIds) + 1, negPerPos, seed,
                                                cand_items=np.asarray(self.productIds), device=self.device)
            X = {"customerId_input": u, "pProduct_input": p, "nProduct_input": n}
        hist = self.fit(X, None, batch_size=self.batchSize, epochs=self.epochs, seed=seed)
        return {"result": "completed", "metrics": [hist.history["loss"][-1]], "history": hist}

    # ---- the evaluation functions of the stand-alone BPR notebook (src/models/bpr.py) on the GPU ----
    def _scores(self, user_ids, items):
        e = self.model
        u = _to_dev(np.asarray(user_ids), e.device, e.id_dtype)
        it = _to_dev(np.asarray(items), e.device, e.id_dtype)
        return e.predict_scores(u, it)                                   # bpr_predict (bpr.py:122-133) for all users at once

    def full_auc(self, ground_truth, items) -> float:
        """full_auc (src/models/bpr.py:230-254): mean over the users that have positives of roc_auc_score(ground truth over all
        `items`, bpr_predict scores).  ground_truth: iterable of (user_id, [true item ids])."""
        gt = list(ground_truth)
        col = {it: j for j, it in enumerate(items)}
        missing = [p for _u, t in gt for p in t if p not in col]
        if missing:      # the reference's `items.index(p)` (bpr.py:247) raises ValueError for a true item outside `items`
            raise ValueError(f"{missing[0]!r} is not in list")
        rows = [r for r, (_u, t) in enumerate(gt) for _ in t]
        cols = [col[p] for _u, t in gt for p in t]
        off, idx = ops.truth_csr(len(gt), rows, cols, self.model.device)
        auc = ops.full_auc(self._scores([u for u, _ in gt], items), off, idx).cpu().numpy()
        has = np.array([len(t) > 0 for _u, t in gt])
        return float(np.mean(auc[has]))

    def mean_average_precision_k(self, ground_truth, items, k=100) -> float:
        """mean_average_precision_k (src/models/bpr.py:257-289): AP of the top-k of the bpr_predict scores per user / min(len(actual), k)."""
        gt = list(ground_truth)
        col = {it: j for j, it in enumerate(items)}
        rows = [r for r, (_u, t) in enumerate(gt) for _ in t]
        cols = [col[p] for _u, t in gt for p in t if p in col]
        rows = [r for r, (_u, t) in enumerate(gt) for p in t if p in col]
        off, idx = ops.truth_csr(len(gt), rows, cols, self.model.device)
        k = min(int(k), len(items))
        _ts, ti = ops.topk_rows(self._scores([u for u, _ in gt], items), k)
        ap, _ = ops.map_at_k(ti, off, idx, want_hits=False)
        # brMapAtK divides by min(truth items it was given, k); the reference by min(len(actual), k) with EVERY listed item, also those
        # outside `items` (bpr.py:286): rescale per user on the host (tiny vectors).  A user without positives: the reference divides by
        # zero there (ZeroDivisionError); here such a user contributes AP 0.
        have = np.array([len({p for p in t if p in col}) for _u, t in gt], dtype=np.float64)
        want = np.array([len(t) for _u, t in gt], dtype=np.float64)
        scale = np.where(want > 0, np.minimum(have, k) / np.maximum(np.minimum(want, k), 1.0), 0.0)
        return float((ap.double().cpu().numpy() * scale).mean())

    def fit(self, X: dict, y=None, batch_size=64, epochs=1, seed=0):
        """model.fit({'customerId_input','pProduct_input','nProduct_input'}, ones, batch_size, epochs) (BPRModel.py:100-109)."""
        e = self.model
        u = _to_dev(X["customerId_input"], e.device, e.id_dtype).view(-1)
        p = _to_dev(X["pProduct_input"], e.device, e.id_dtype).view(-1)
        n = _to_dev(X["nProduct_input"], e.device, e.id_dtype).view(-1)
        hist = History()
        g = torch.Generator(device=e.device).manual_seed(seed)
        ctx = getattr(e, "ctx", None)                 # row-sharded engine (compileModel under a process group): this rank's slice of each global batch
        slices = _rank_slices(u.shape[0], batch_size * (ctx.world if ctx is not None else 1), ctx)
        for _ in range(epochs):
            perm = torch.randperm(u.shape[0], device=e.device, generator=g)
            uu, pp, nn = u[perm], p[perm], n[perm]
            for lo, hi, _row0, bt in slices:
                e.train_step(uu[lo:hi], pp[lo:hi], nn[lo:hi], batch_total=bt)
            e.check_ids()
            if ctx is not None:
                ctx.all_reduce_sum(e.loss_slots)      # the epoch's loss over all replicas (a collective)
                e.n_seen = sum(s[3] for s in slices) * 1
            hist.add({"loss": e.pop_loss()})
        return hist


class TwoTowerModel:
    """trainers/twoTower.py:19-111 with the same constructor arguments."""

    def __init__(self, embedDim, nbrItem, nbrUser, userKey, itemKey, usersId, itemsId, eval_batch_size=8000, loss=None,
                 rdZero=False, resKey=None, semb=100, device="cuda:0", max_batch=65536, learningRate=0.1, optimiser="Adagrad"):
        self.userKey, self.itemKey, self.resKey, self.rdZero, self.eval_batch_size = userKey, itemKey, resKey, rdZero, eval_batch_size
        self.userTowerIn, self.itemTowerIn = StringLookup(usersId), StringLookup(itemsId)
        self.engine = TwoTowerEngine(embedDim, nbrItem, nbrUser, semb, device, max_batch, lr=learningRate, optimizer=optimiser, rd_zero=rdZero)
        self.device = self.engine.device
        self._cand, self._cand_ids, self._k = None, None, None
        self._loss_seen = torch.zeros((), dtype=torch.float64, device=self.device)

    def _ids(self, info):
        return (self.userTowerIn(info[self.userKey], self.device), self.itemTowerIn(info[self.itemKey], self.device))

    def computeEmb(self, info):
        u, i = self._ids(info)
        return self.engine.compute_emb(u, i, u.shape[0])

    def computeLossTfrs(self, usersCaracteristics, itemCaracteristics, info):
        """trainers/twoTower.py:82-83: self.task(q, c, compute_metrics=False, training=True, candidate_ids=info[itemKey]) with
        task = tfrs.tasks.Retrieval(loss=None) [TF-sem, TFRS unpinned]: in-batch softmax over the batch's candidates, accidental hits
        (another position with the SAME item id) masked with float32 min / 100, categorical cross-entropy from logits, reduction SUM.
        -> the loss of (q, c) as a 0-dim float32 DEVICE tensor (no host sync), by brInBatchSoftmaxLse."""
        q, c = _to_dev(usersCaracteristics, self.device, torch.float32), _to_dev(itemCaracteristics, self.device, torch.float32)
        ids = self.itemTowerIn(info[self.itemKey], self.device)
        slots = torch.zeros(ops.SUM_SLOTS, dtype=torch.float64, device=self.device)
        lse = torch.empty(q.shape[0], dtype=torch.float32, device=self.device)
        ops.inbatch_softmax_lse(q, c, ids, ids, 0, lse, slots)
        return slots.sum().float()

    def computeLossRdZero(self, usersCaracteristics, itemCaracteristics, info):
        """trainers/twoTower.py:85-87: compiled_loss(info[resKey], sigmoid(Dot(axes=-1)([q, c]))) with the BinaryCrossentropy the trainer
        compiles (twoTower.py:209): mean over the batch, evaluated from the logit (brRowDot + brBceLogits) -> 0-dim float32 device tensor."""
        q, c = _to_dev(usersCaracteristics, self.device, torch.float32), _to_dev(itemCaracteristics, self.device, torch.float32)
        n = q.shape[0]
        y = _to_dev(info[self.resKey], self.device, torch.float32).view(-1)
        z = ops.row_dot(q, c)
        slots = torch.zeros(ops.SUM_SLOTS, dtype=torch.float64, device=self.device)
        ops.bce_logits(z, y, 1.0 / n, sums=slots)
        return (slots.sum() / n).float()

    def computeLoss(self, usersCaracteristics, itemCaracteristics, info):
        """bound at construction like the reference's `self.computeLoss = ...` (twoTower.py:45,48)"""
        return (self.computeLossRdZero if self.rdZero else self.computeLossTfrs)(usersCaracteristics, itemCaracteristics, info)

    def _step_loss(self, batch, mean):
        """the step's loss as a 0-dim DEVICE tensor (metrics["loss"] = loss, twoTower.py:99-102,107-111): the difference of the
        engine's running loss sum across the step (softmax: the TFRS SUM over the batch; rdZero train_step: mean BCE, as the
        compiled loss; test_step: always the retrieval task's SUM, twoTower.py:106) - no host sync; float() / .item() on it is the
        caller's.  The running sum only grows (both losses are >= 0), so a total BELOW the last one seen means somebody read and
        cleared the slots in between (engine.pop_loss()): the step's loss is then the total itself."""
        total = self.engine.loss_slots.sum()
        loss = torch.where(total < self._loss_seen, total, total - self._loss_seen)
        self._loss_seen = total
        return loss / batch if mean else loss

    def train_step(self, info):
        u, i = self._ids(info)
        y = None if not self.rdZero else _to_dev(info[self.resKey], self.device, torch.float32)
        self.engine.train_step(u, i, y)
        return {"loss": self._step_loss(u.shape[0], self.rdZero)}

    def test_step(self, info):
        u, i = self._ids(info)
        self.engine.test_step(u, i)
        return {"loss": self._step_loss(u.shape[0], False)}

    def fit(self, batches, epochs=1):
        """model.fit(trainSetCached, epochs) (twoTower.py:214): `batches` = iterable of info dicts."""
        hist = History()
        for _ in range(epochs):
            for info in batches:
                self.train_step(info)
            self.engine.check_ids()
            hist.add({"loss": self.engine.pop_loss()})
            self._loss_seen.zero_()          # pop_loss cleared the running sum
        return hist

    def setCandidates(self, items, k):
        """BruteForce(k).index(itemTower(items), identifiers=items) (twoTower.py:64-69)."""
        self._cand_ids = list(items)
        self._cand = self.engine.item_tower(self.itemTowerIn(self._cand_ids, self.device))
        self._k = k

    def call(self, users):
        """userTower -> BruteForce top-k (twoTower.py:60-62) -> (scores (U,k), identifiers (U,k))."""
        q = self.engine.user_tower(self.userTowerIn(users, self.device))
        ts, ti = ops.topk_rows(ops.score_matrix(q, self._cand), self._k)
        ids = np.asarray(self._cand_ids, dtype=object)[ti.cpu().numpy()]
        return ts.cpu().numpy(), ids

    predict = call

    def topk(self, usersId, itemsId, k):
        self.setCandidates(itemsId, k)
        q = self.engine.user_tower(self.userTowerIn(usersId, self.device))
        return ops.topk_rows(ops.score_matrix(q, self._cand), k)
