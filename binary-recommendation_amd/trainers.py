"""trainers/twoTower.py's 5-fold cross-validation driver on the HIP path — crossValidation, twoTower.py:125-272.

The reference loads its folds from an SMB share (input() / getpass(), :130-139: out of scope) and feeds tf.data pipelines; here
the folds are handed in as in-memory tables ({user key: [...], item key: [...], "RATING_TYPE": [...]} dicts or DataFrames) and the
loop is the same: hold one fold out, train a fresh TwoTowerModel on the others with the utilisation sampler running
(benchThread, :202-223), index the candidates (setCandidates, :229), predict the top-k of every user (:230), topKMetrics against
the held-out fold (:241) and against the whole data (:248-251), average over the folds (:256-272)."""
from __future__ import annotations

import os
from datetime import datetime

import numpy as np

from .benchmark_logger import benchThread
from .models import TwoTowerModel
from .topk_metrics import topKMetrics


def _col(ds, key):
    v = ds[key]
    return v.tolist() if hasattr(v, "tolist") else list(v)


def _batches(ds, userKey, itemKey, resKey, batchSize, order):
    u, i = _col(ds, userKey), _col(ds, itemKey)
    r = _col(ds, resKey) if resKey in ds else None
    out = []
    for s in range(0, len(order), batchSize):
        idx = order[s:s + batchSize]
        b = {userKey: [u[j] for j in idx], itemKey: [i[j] for j in idx]}
        if r is not None:
            b[resKey] = [r[j] for j in idx]
        out.append(b)
    return out


def _concat(dss, keys):
    return {k: [x for ds in dss for x in _col(ds, k)] for k in keys if all(k in ds for ds in dss)}


def crossValidation(dataSets, k, learningRate, optimiser, loss, epoch, embNum, batchSize, randomZero=False, rdZeroDataSets=None,
                    testBatchSize=5000, semb=64, bname=None, userKey="CUSTOMER_ID", itemKey="MATERIAL", resKey="RATING_TYPE",
                    device="cuda:0", seed=0, pollTime=0.01):
    """-> averaged metrics dict: {tp, tn, fp, fn, precision, recall, hitRate, full_*} (twoTower.py:256-272).
    dataSets: the folds (positives, RATING_TYPE 1); rdZeroDataSets: the same folds with randomly added zeros, trained on when
    randomZero (the "rdZero" sigmoid-BCE loss, twoTower.py:85-87,176-177) while the plain folds stay the test sets."""
    dataSets = list(dataSets)
    keys = (userKey, itemKey, resKey)
    usersId = list(dict.fromkeys(str(x) for ds in dataSets for x in _col(ds, userKey)))
    matId = list(dict.fromkeys(str(x) for ds in dataSets for x in _col(ds, itemKey)))
    if bname:
        os.makedirs(bname, exist_ok=True)
    train_sets = list(rdZeroDataSets) if randomZero else dataSets
    rng = np.random.default_rng(seed)
    res, fullRes = [], []
    n_folds = len(dataSets)
    for it in range(n_folds):
        testData = dataSets[it]
        trainSet = _concat([ds for j, ds in enumerate(train_sets) if j != it], keys)
        order = rng.permutation(len(trainSet[userKey])).tolist()          # shuffle(len, reshuffle_each_iteration=False) (:197)
        batches = _batches(trainSet, userKey, itemKey, resKey, batchSize, order)
        bm = None
        if bname:
            bm = benchThread(pollTime, 1, os.path.join(bname, "it" + str(it) + "_" + datetime.now().strftime("%d_%H_%M_%S")))
            bm.start()
        try:
            model = TwoTowerModel(embNum, len(matId), len(usersId), userKey, itemKey, usersId, matId, eval_batch_size=batchSize, loss=loss,
                                  rdZero=randomZero, resKey=resKey, semb=semb, device=device, max_batch=batchSize, learningRate=learningRate,
                                  optimiser=optimiser)
            model.fit(batches, epochs=epoch)
        finally:
            if bm is not None:
                bm.active = 0
                bm.join()
        model.setCandidates(matId, k)
        scores, ids = model.predict(usersId)
        topk = [(str(u), [(float(scores[n][j]), str(ids[n][j])) for j in range(scores.shape[1])]) for n, u in enumerate(usersId)]
        pairs = lambda ds: [(str(a), str(b)) for a, b in zip(_col(ds, userKey), _col(ds, itemKey))]
        res.append(topKMetrics(topk, pairs(testData), usersId, matId))
        fullRes.append(topKMetrics(topk, [p for ds in dataSets for p in pairs(ds)], usersId, matId))
    avg = {m: sum(r[m] for r in res) / n_folds for m in res[0]}
    avg.update({"full_" + m: sum(r[m] for r in fullRes) / n_folds for m in fullRes[0]})
    return avg
