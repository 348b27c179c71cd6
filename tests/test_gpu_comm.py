"""GPU: the RCCL path behind the boundary (csrc/comm.hip: gorse_comm_*, gorse_mf_item_allreduce, gorse_mf_rows_allgather,
gorse_comm_allreduce_f32) on what a one-GPU box can run of it -- communicators of ONE rank, created both ways (unique id +
gorse_comm_create, gorse_comm_create_local).  librccl is opened, the collectives run on the handle's stream between the
export and import kernels, and the results equal the exchange arithmetic done on the host.  The multi-rank arithmetic
itself is covered by tests/test_dist_gloo.py (world 2 and 3, same run_epoch / run_als_epoch functions)."""
import numpy as np
import pytest

from gorse_amd import capi, synth
from gorse_amd import dist as gdist

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def data():
    return synth.synth_cf(5000, 700, 60000, seed=3, min_len=3, n_neg=10)


@pytest.mark.parametrize("how", ["unique_id", "local"])
def test_item_allreduce_world_1(data, how):
    d = 64
    comm = capi.Comm(capi.Comm.unique_id(), 1, 0, 0) if how == "unique_id" else capi.Comm.local([0])[0]
    assert (comm.world, comm.rank) == (1, 0)
    mf = capi.MF(data.U, data.I, d, data.uptr, data.uidx)
    P0, Q0 = synth.init_factors(data.U, data.I, d, 0.0, 0.1, 1)
    mf.set_factors(P0, Q0)
    mf.item_sync_mark()
    mf.bpr_epoch(data.n_train, 0.05, 0.01, 11, 1, mode=capi.BPR_HOGWILD_ATOMIC)
    P1, Q1 = mf.get_factors()
    mf.set_profiling(True)
    capi.item_allreduce([mf], [comm])  # Q <- Q_sync + sum over the one rank of (Q - Q_sync)
    P2, Q2 = mf.get_factors()
    n, ms = mf.get_profile(capi.PROF_COMM)
    assert n == 1 and ms > 0
    expect = (Q0 + (Q1 - Q0)).astype(np.float32)
    assert np.array_equal(Q2.view(np.uint32), expect.view(np.uint32)) and np.array_equal(P2.view(np.uint32), P1.view(np.uint32))
    capi.item_allreduce([mf], [comm])  # Q_sync is Q now: the delta is zero
    assert np.array_equal(mf.get_factors()[1].view(np.uint32), Q2.view(np.uint32))
    comm.close()


def test_rows_allgather_and_host_allreduce_world_1(data):
    d = 16
    comm = capi.Comm(capi.Comm.unique_id(), 1, 0, 0)
    mf = capi.MF(data.U, data.I, d, data.uptr, data.uidx, data.iptr, data.iidx)
    P0, Q0 = synth.init_factors(data.U, data.I, d, 0.0, 0.1, 2)
    mf.set_factors(P0, Q0)
    mf.als_half_epoch(0, 0.05, 0.015)
    P1, _ = mf.get_factors()
    capi.rows_allgather([mf], [comm], 0, [0, data.U])
    capi.rows_allgather([mf], [comm], 1, [0, data.I])
    P2, Q2 = mf.get_factors()
    assert np.array_equal(P2.view(np.uint32), P1.view(np.uint32)) and np.array_equal(Q2.view(np.uint32), Q0.view(np.uint32))
    with pytest.raises(capi.GorseHipError) as e:
        capi.rows_allgather([mf], [comm], 0, [0, data.U - 1])
    assert e.value.code == capi.ERR_RANGE
    assert comm.allreduce_f32([1.5, -2.0, 3.25]).tolist() == [1.5, -2.0, 3.25]
    assert capi.comm_available() is None
    got = capi.allreduce_f32_local(capi.Comm.local([0]), [np.array([0.5, 4.0], np.float32)])  # the one-process form, a group of one
    assert got[0].tolist() == [0.5, 4.0]


def test_run_epoch_through_the_library_communicator(data):
    """gorse_amd.dist.run_epoch / run_als_epoch / evaluate_sharded with a LibComm (what bench.py --gpus N uses): one rank"""
    d = 32
    comm = gdist.LibComm(0, 1, 0, share=lambda b: b, always=True)
    mf = capi.MF(data.U, data.I, d, data.uptr, data.uidx, data.iptr, data.iidx)
    P0, Q0 = synth.init_factors(data.U, data.I, d, 0.0, 0.1, 4)
    mf.set_factors(P0, Q0)
    mf.item_sync_mark()
    eng = gdist.HipEngine(mf, capi.BPR_HOGWILD_ATOMIC)
    for ep in (1, 2):
        gdist.run_epoch(eng, comm, data.n_train, 0.05, 0.01, 5, ep, 0)
    P, Q = mf.get_factors()
    assert np.isfinite(P).all() and np.isfinite(Q).all() and not np.array_equal(Q, Q0)
    eng.set_eval(data.test_ptr, data.test_idx, data.neg_ptr, data.neg_idx)
    got = gdist.evaluate_sharded(eng, comm, 10)
    ref = gdist.evaluate_sharded(eng, None, 10)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    als = gdist.HipAlsEngine(mf, 0, 1)
    gdist.run_als_epoch(als, comm, 0.05, 0.015)
    ref_mf = capi.MF(data.U, data.I, d, data.uptr, data.uidx, data.iptr, data.iidx)
    ref_mf.set_factors(P, Q)
    ref_mf.als_epoch(0.05, 0.015)
    for x, y in zip(mf.get_factors(), ref_mf.get_factors()):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    comm.close()


def test_one_process_local_communicators_call_sequence(data):
    """run_epoch_local / run_als_epoch_local (the Go master's order: every device's kernels enqueued, then ONE grouped exchange over
    all (handle, communicator) pairs of the process) on what a one-GPU box holds of it: a group of one.  The N > 1 arithmetic
    of the same two functions is covered by tests/test_dist_gloo.py::test_one_process_n_handles_call_sequence."""
    d = 32
    comms = gdist.LocalComms([0])
    mf = capi.MF(data.U, data.I, d, data.uptr, data.uidx, data.iptr, data.iidx)
    P0, Q0 = synth.init_factors(data.U, data.I, d, 0.0, 0.1, 4)
    mf.set_factors(P0, Q0)
    mf.item_sync_mark()
    eng = gdist.HipEngine(mf, capi.BPR_HOGWILD_ATOMIC)
    gdist.run_epoch_local([eng], comms, [data.n_train], 0.05, 0.01, 5, 1)
    P, Q = mf.get_factors()
    assert np.isfinite(P).all() and np.isfinite(Q).all() and not np.array_equal(Q, Q0)
    als = gdist.HipAlsEngine(mf, 0, 1, staging=False)
    gdist.run_als_epoch_local([als], comms, 0.05, 0.015)  # enqueued half-sweeps + grouped all-gathers, one synchronisation
    ref_mf = capi.MF(data.U, data.I, d, data.uptr, data.uidx, data.iptr, data.iidx)
    ref_mf.set_factors(P, Q)
    ref_mf.als_epoch(0.05, 0.015)
    for x, y in zip(mf.get_factors(), ref_mf.get_factors()):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    comms.close()
