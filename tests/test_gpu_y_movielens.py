"""GPU: the reference's own end-to-end accuracy tests (model/cf/model_test.go:35-122, TestBPR_MovieLens / TestCCD_MovieLens)
on the REAL ml-1m files -- only where they exist (~/.gorse/dataset/ml-1m/{train,test}.txt in the NCF layout the reference
downloads, dataset/dataset.go:398-490); the build image has no network, so here they are skipped and the synthetic S-ml1m
stand-ins of tests/test_gpu_cf_parity.py carry the NDCG parity.  Listed in tests/conftest.py ISOLATED_GPU_MODULES."""
import os

import pytest

from gorse_amd import cf

pytestmark = pytest.mark.gpu
DIR = os.path.expanduser("~/.gorse/dataset/ml-1m")
FILES = [os.path.join(DIR, f) for f in ("train.txt", "test.txt")]
needs_files = pytest.mark.skipif(not all(os.path.exists(f) for f in FILES), reason="the real ml-1m files are not on this machine")


def _load():
    train, test = cf.Dataset.LoadNCF(open(FILES[0]).read(), open(FILES[1]).read())
    assert (train.CountUsers(), train.CountItems(), train.CountFeedback()) == (6040, 3706, 994169)  # dataset_test.go:264-272
    assert test.CountFeedback() == 6040
    return train, test


def _config():
    c = cf.FitConfig()
    c.Verbose, c.Jobs = 1, os.cpu_count() or 1  # newFitConfig (model_test.go:30-33)
    return c


def _structure(m):
    assert m.Predict("1", "1") == m.internalPredict(1, 1)
    assert m.IsUserPredictable(1) and m.IsItemPredictable(1)
    assert not m.IsUserPredictable(2 ** 31 - 1) and not m.IsItemPredictable(2 ** 31 - 1)
    tmp = cf.UnmarshalModel(cf.MarshalModel(m))
    assert tmp.Predict("1", "1") == m.Predict("1", "1")
    m.Clear()
    assert m.Invalid()


@needs_files
def test_bpr_movielens():
    train, test = _load()
    m = cf.BPR({"NFactors": 8, "Reg": 0.01, "Lr": 0.05, "NEpochs": 30, "InitMean": 0, "InitStdDev": 0.001})
    score = m.Fit(train, test, _config())
    assert abs(score.NDCG - 0.36) < 0.01  # benchDelta
    _structure(m)


@needs_files
def test_ccd_movielens():
    train, test = _load()
    m = cf.ALS({"NFactors": 8, "Reg": 0.015, "NEpochs": 30, "Alpha": 0.05})
    score = m.Fit(train, test, _config())
    assert abs(score.NDCG - 0.36) < 0.01
    _structure(m)
