"""The chromosome-sharded pipeline (canvas_sample_pipeline_sharded) must return, on every rank, exactly what the single-GPU pipeline returns.
Two processes share the one GPU of the test box, so they cannot form an RCCL communicator: their exchanges go through the library's host-callback transport
(torch.distributed / gloo underneath); the RCCL transport is exercised with a one-rank communicator in test_single_rank_rccl_communicator."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LENGTHS = [1_400_000, 1_100_003, 900_000, 650_000, 420_000, 300_001, 90_000]
SEED = 20260927 + 4


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _inputs(device, only=None):
    import torch
    from canvas_amd import synth
    thr = synth.poisson_thresholds(0.21)
    pad = lambda a: np.concatenate([a, np.zeros((-len(a)) % 64, a.dtype)])
    bases, hits, masks = [], [], []
    for c, L in enumerate(LENGTHS):
        if only is not None and c not in only:
            bases.append(None); hits.append(None); masks.append(None); continue
        b, h, m = synth.generate_chromosome(SEED, c, L, 0.21, thr)
        if c in (0, 2, 5):                                      # a heterozygous deletion: every other hit of a stretch removed, so that the HMM has segments to find
            a0, a1 = L // 4, L // 2
            h = h.copy(); h[a0:a1] = np.where(np.arange(a0, a1) % 2 == 0, h[a0:a1], 0)
        bases.append(torch.from_numpy(pad(b)).to(device)); hits.append(torch.from_numpy(pad(h)).to(device)); masks.append(torch.from_numpy(m.view(np.int64).copy()).to(device))
    return bases, hits, masks


def _buffers(device):
    import torch
    cap = sum(LENGTHS) // 100 + 64
    mk = lambda dt: torch.empty(cap, dtype=dt, device=device)
    return dict(chr=mk(torch.int32), start=mk(torch.int32), stop=mk(torch.int32), gc=mk(torch.int32), count=mk(torch.float32)), mk(torch.float64), mk(torch.int32), mk(torch.int32)


FLAGS = 1 | 2 | 4 | 8
IS_AUTO = np.array([1, 1, 1, 1, 1, 0, 0], np.uint8)


def _single(cv):
    bases, hits, masks = _inputs(cv.device)
    out, cov, state, seg = _buffers(cv.device)
    r = cv.sample_pipeline(bases, masks, hits, np.array(LENGTHS, np.int64), IS_AUTO, out, cov, state, seg, counts_per_bin=100, bin_size=-1, mode=3, flags=FLAGS)
    cv.synchronize()
    n = r["n_out"]
    return r, {k: v[:n].cpu().numpy() for k, v in out.items()}, cov[:n].cpu().numpy(), state[:n].cpu().numpy(), seg[:n].cpu().numpy()


def _worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from canvas_amd import Canvas, parallel
        cv = Canvas(0)
        parallel.init_host_comm(cv, rank, world)
        owner = parallel.owner_table(LENGTHS, world)
        mine = [c for c in range(len(LENGTHS)) if owner[c] == rank]
        bases, hits, masks = _inputs(cv.device, only=mine)
        out, cov, state, seg = _buffers(cv.device)
        res = []
        for bin_size in (-1, 250):                              # derived from the exchanged rates / given (-z)
            r = cv.sample_pipeline_sharded(owner, bases, masks, hits, np.array(LENGTHS, np.int64), IS_AUTO, out, cov, state, seg, counts_per_bin=100, bin_size=bin_size, mode=3, flags=FLAGS)
            cv.synchronize()
            n = r["n_out"]
            res.append((dict(r, off=r["off"].tolist()), {k: v[:n].cpu().numpy() for k, v in out.items()}, cov[:n].cpu().numpy(), state[:n].cpu().numpy(), seg[:n].cpu().numpy(),
                        cv.sharded_stats().tolist()))
        # the same two passes over the packed planes of the owned chromosomes (canvas_sample_pipeline_sharded_packed): results are appended, the test expects them to repeat
        lens_mine = [LENGTHS[c] for c in mine]
        ref_m, planes_m, pos0_m, _ = cv.pack_genome_device([bases[c] for c in mine], [masks[c] for c in mine], [hits[c] for c in mine], lens_mine)
        ref, planes, pos0 = [None] * len(LENGTHS), [None] * len(LENGTHS), np.zeros(len(LENGTHS), np.int64)
        for i, c in enumerate(mine):
            ref[c], planes[c], pos0[c] = ref_m[i], planes_m[i], pos0_m[i]
        for bin_size in (-1, 250):
            r = cv.sample_pipeline_sharded(owner, ref, None, planes, np.array(LENGTHS, np.int64), IS_AUTO, out, cov, state, seg, counts_per_bin=100, bin_size=bin_size, mode=3, flags=FLAGS, pos0=pos0)
            cv.synchronize()
            n = r["n_out"]
            res.append((dict(r, off=r["off"].tolist()), {k: v[:n].cpu().numpy() for k, v in out.items()}, cov[:n].cpu().numpy(), state[:n].cpu().numpy(), seg[:n].cpu().numpy(),
                        cv.sharded_stats().tolist()))
        q.put((rank, owner.tolist(), res))
        dist.destroy_process_group()
    except Exception as e:                                      # noqa: BLE001
        import traceback
        q.put((rank, "error", traceback.format_exc()))


def test_two_ranks_on_one_gpu_equal_the_single_rank_result():
    import torch
    import torch.multiprocessing as mp
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    got = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs: p.join(60)
    for g in got:
        assert g[1] != "error", g[2]
    from canvas_amd import Canvas
    cv = Canvas(0)
    owner = got[0][1]
    assert sorted(set(owner)) == [0, 1]
    for k, bin_size in enumerate((-1, 250)):
        if bin_size == -1:
            ref = _single(cv)
        else:
            bases, hits, masks = _inputs(cv.device)
            out, cov, state, seg = _buffers(cv.device)
            r = cv.sample_pipeline(bases, masks, hits, np.array(LENGTHS, np.int64), IS_AUTO, out, cov, state, seg, counts_per_bin=100, bin_size=bin_size, mode=3, flags=FLAGS)
            cv.synchronize(); n = r["n_out"]
            ref = (r, {kk: v[:n].cpu().numpy() for kk, v in out.items()}, cov[:n].cpu().numpy(), state[:n].cpu().numpy(), seg[:n].cpu().numpy())
        r1 = ref[0]
        for rank, _, res in [(g[0], g[1], g[2]) for g in got] + [(g[0], g[1], g[2][2:]) for g in got]:      # byte arrays, then the packed planes
            r, o, cv_, st, sg, stats = res[k]
            assert (r["bin_size"], r["total"], r["n_out"], r["nseg"], r["lsd"]) == (r1["bin_size"], r1["total"], r1["n_out"], r1["nseg"], r1["lsd"]), (rank, bin_size)
            assert r["off"] == r1["off"].tolist()
            for key in ("chr", "start", "stop", "gc"):
                assert (o[key] == ref[1][key]).all(), (rank, key)
            assert (o["count"].view(np.uint32) == ref[1]["count"].view(np.uint32)).all()
            assert (cv_ == ref[2]).all() and (st == ref[3]).all() and (sg == ref[4]).all()
            assert stats[0] == 2 and stats[1] == owner.count(rank) and stats[4] > 0
        assert r1["nseg"] > len(LENGTHS)                      # the planted copy-number segments are there


def test_single_rank_rccl_communicator():
    """the RCCL transport (ncclAllGather on the library's stream) with a communicator of one rank: same code path as N ranks, same result as the plain pipeline"""
    import ctypes as C
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from canvas_amd import Canvas
    cv = Canvas(0)
    buf = (C.c_ubyte * 128)()
    cv._check(cv.lib.canvas_comm_unique_id(buf))
    cv._check(cv.lib.canvas_comm_init(cv.ctx, 0, 1, buf))
    ref = _single(cv)
    bases, hits, masks = _inputs(cv.device)
    out, cov, state, seg = _buffers(cv.device)
    r = cv.sample_pipeline_sharded(np.zeros(len(LENGTHS), np.int32), bases, masks, hits, np.array(LENGTHS, np.int64), IS_AUTO, out, cov, state, seg, counts_per_bin=100, bin_size=-1, mode=3, flags=FLAGS)
    cv.synchronize()
    n = r["n_out"]
    assert (r["total"], n, r["nseg"]) == (ref[0]["total"], ref[0]["n_out"], ref[0]["nseg"])
    assert (seg[:n].cpu().numpy() == ref[4]).all() and (state[:n].cpu().numpy() == ref[3]).all()
    assert (out["count"][:n].cpu().numpy().view(np.uint32) == ref[1]["count"].view(np.uint32)).all()
    # the boundary all-gather itself, as the ABI exposes it
    rec = torch.tensor([3, 0, 9, 2, 3, 10, 40, 1], dtype=torch.int32, device=cv.device)
    allb = torch.zeros(1 + 16, dtype=torch.int32, device=cv.device)
    cnt = np.zeros(1, np.int32)
    torch.cuda.synchronize()
    cv._check(cv.lib.canvas_allgather_boundaries(cv.ctx, C.c_void_p(rec.data_ptr()), 8, 16, C.c_void_p(allb.data_ptr()), cnt.ctypes.data_as(C.c_void_p)))
    assert cnt[0] == 8 and allb[:9].cpu().tolist() == [8, 3, 0, 9, 2, 3, 10, 40, 1]


def _failing_worker(rank, world, port, q, where):
    """rank 1 fails on its own — before the first exchange (an owned chromosome without arrays) or between the exchanges (an output capacity only it finds too small);
    the library must bring BOTH ranks back with an error instead of leaving rank 0 in an all-gather"""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from canvas_amd import Canvas, parallel
        from canvas_amd.lib import CanvasError
        cv = Canvas(0)
        parallel.init_host_comm(cv, rank, world)
        owner = parallel.owner_table(LENGTHS, world)
        mine = [c for c in range(len(LENGTHS)) if owner[c] == rank]
        bases, hits, masks = _inputs(cv.device, only=mine)
        out, cov, state, seg = _buffers(cv.device)
        if rank == 1 and where == "before":
            hits[mine[0]] = None
        if rank == 1 and where == "between":
            out = {k: v[:1000] for k, v in out.items()}          # far fewer rows than the sample has bins: CANVAS_ERR_CAPACITY on this rank only
        try:
            cv.sample_pipeline_sharded(owner, bases, masks, hits, np.array(LENGTHS, np.int64), IS_AUTO, out, cov, state, seg, counts_per_bin=100, bin_size=-1, mode=3, flags=FLAGS)
            q.put((rank, "returned", ""))
        except CanvasError as e:
            q.put((rank, "raised", str(e)))
        dist.destroy_process_group()
    except Exception:                                           # noqa: BLE001
        import traceback
        q.put((rank, "error", traceback.format_exc()))


@pytest.mark.parametrize("where", ["before", "between"])
def test_a_failure_on_one_rank_fails_every_rank_instead_of_deadlocking(where):
    import torch
    import torch.multiprocessing as mp
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, world, port, q, where)) for r in range(world)]
    for p in procs: p.start()
    try:
        got = dict((r, (what, msg)) for r, what, msg in [q.get(timeout=180) for _ in range(world)])
    finally:
        for p in procs:
            p.join(20)
            if p.is_alive(): p.kill()
    assert got[0][0] == "raised" and got[1][0] == "raised", got
    assert "rank 1 failed" in got[0][1], got[0][1]            # the healthy rank names the one that failed
    assert ("no arrays" in got[1][1]) if where == "before" else ("capacity" in got[1][1]), got[1][1]


def _rccl_worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(rank)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from canvas_amd import Canvas, parallel
        cv = Canvas(rank)
        parallel.init_library_comm(cv, rank, world)
        owner = parallel.owner_table(LENGTHS, world)
        mine = [c for c in range(len(LENGTHS)) if owner[c] == rank]
        bases, hits, masks = _inputs(cv.device, only=mine)
        out, cov, state, seg = _buffers(cv.device)
        r = cv.sample_pipeline_sharded(owner, bases, masks, hits, np.array(LENGTHS, np.int64), IS_AUTO, out, cov, state, seg, counts_per_bin=100, bin_size=-1, mode=3, flags=FLAGS)
        cv.synchronize()
        n = r["n_out"]
        # the sharded partition methods on the coverage the pipeline left on every rank
        seg_len, nseg_c, _ = cv.cbs_sharded(owner, cov, r["off"], 0.01, 500)
        cbs = [seg_len.cpu().numpy()[int(r["off"][c]):int(r["off"][c]) + int(nseg_c[c])].tolist() for c in range(len(LENGTHS))]
        wv = [b.tolist() for b in cv.wavelets_sharded(owner, cov, r["off"], window=2000)]
        # ---- sub-communicators of two colors (even / odd ranks): the group's rank and size, an exchange inside the group, the parent back afterwards
        gr, gn = parallel.split_library_comm(cv, rank % 2, rank)
        members = [x for x in range(world) if x % 2 == rank % 2]
        inside = cv.allgather_host(np.array([rank], np.int64)).ravel().tolist()
        pr, pn = parallel.restore_library_comm(cv)
        split = dict(group_rank=gr, group_size=gn, expected=(members.index(rank), len(members)), gathered=inside, members=members, parent=(pr, pn))
        # ---- CanvasBin -m GCContentWeighted with the chromosomes sharded (two reductions + the rate table + the bins over RCCL)
        gowner = parallel.owner_table(GCW_LENGTHS, world)
        gb, gh, gm, gf = _gcw_inputs(cv.device, only=[c for c in range(len(GCW_LENGTHS)) if gowner[c] == rank])
        gout = _bins_out(cv.device)
        gbs, gtotal = cv.bin_sample_sharded(gowner, gb, gm, gh, np.array(GCW_LENGTHS, np.int64), GCW_AUTO, gout, counts_per_bin=100, bin_size=-1, mode=5, fraglens=gf)
        gcw = (gbs, gtotal, {k: v[:gtotal].cpu().numpy() for k, v in gout.items()})
        q.put((rank, dict(r, off=r["off"].tolist()), out["count"][:n].cpu().numpy(), state[:n].cpu().numpy(), seg[:n].cpu().numpy(), cbs, wv, split, gcw))
        dist.destroy_process_group()
    except Exception:                                           # noqa: BLE001
        import traceback
        q.put((rank, "error", traceback.format_exc()))


def test_rccl_all_gather_between_real_ranks():
    """ncclAllGather over xGMI with one process per GPU: runs wherever the box has at least two GPUs (the single-GPU test boxes skip it)"""
    import torch
    import torch.multiprocessing as mp
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs at least two GPUs")
    world = min(torch.cuda.device_count(), 8)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    try:
        got = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    finally:
        for p in procs:
            p.join(60)
            if p.is_alive(): p.kill()
    for g in got:
        assert g[1] != "error", g[2]
    from canvas_amd import Canvas
    cv0 = Canvas(0)
    ref = _single(cv0)
    dcov = torch.from_numpy(ref[2]).to(cv0.device); off = ref[0]["off"]
    seg_len, nseg_c, _ = cv0.cbs(dcov, off, 0.01, 500)
    cbs1 = [seg_len.cpu().numpy()[int(off[c]):int(off[c]) + int(nseg_c[c])].tolist() for c in range(len(LENGTHS))]
    wv1 = [b.tolist() for b in cv0.wavelets(dcov, off, window=2000)]
    gb, gh, gm, gf = _gcw_inputs(cv0.device)
    g1 = _bins_out(cv0.device)
    _, _, gtot1, gbs1 = cv0.bin_sample_gcweighted(gb, gm, gh, gf, np.array(GCW_LENGTHS, np.int64), GCW_AUTO, 100, -1, out=g1)
    for rank, r, count, state, seg, cbs, wv, split, gcw in got:
        assert (r["bin_size"], r["total"], r["n_out"], r["nseg"], r["lsd"]) == (ref[0]["bin_size"], ref[0]["total"], ref[0]["n_out"], ref[0]["nseg"], ref[0]["lsd"]), rank
        assert (count.view(np.uint32) == ref[1]["count"].view(np.uint32)).all() and (state == ref[3]).all() and (seg == ref[4]).all(), rank
        assert cbs == cbs1 and wv == wv1, rank
        assert (split["group_rank"], split["group_size"]) == split["expected"] and split["gathered"] == split["members"] and split["parent"] == (rank, world), (rank, split)
        assert (gcw[0], gcw[1]) == (gbs1, gtot1) and all((gcw[2][k].view(np.uint32) == g1[k][:gtot1].cpu().numpy().view(np.uint32)).all() for k in g1), rank


# ---- CanvasPartition -m CBS / -m Wavelets with the chromosomes sharded over the ranks (canvas_cbs_sharded, canvas_wavelets_sharded)
PART_LENS = [6000, 4100, 3000, 2500, 1200, 700, 9]            # the last chromosome is below Wavelets' MinSize


def _partition_coverage():
    rng = np.random.default_rng(SEED + 11)
    parts = []
    for c, L in enumerate(PART_LENS):
        x = rng.normal(60.0, 6.0, L)
        for k in range(1 + c % 3):                              # planted copy-number changes
            a = int(rng.integers(0, max(1, L - 50))); b = min(L, a + int(rng.integers(30, max(31, L // 3))))
            x[a:b] *= rng.choice([0.5, 1.5, 2.0])
        parts.append(np.round(np.clip(x, 0, None), 2))
    cov = np.concatenate(parts)
    off = np.concatenate([[0], np.cumsum(PART_LENS)]).astype(np.int64)
    return cov, off


def _partition_worker(rank, world, port, q, fail, list_first=None):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        if list_first: os.environ["CANVAS_SHARDED_LIST_FIRST"] = str(list_first)       # the list exchange starts with a size the lists overflow: every rank steps up together
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from canvas_amd import Canvas, parallel
        from canvas_amd.lib import CanvasError
        cv = Canvas(0)
        parallel.init_host_comm(cv, rank, world)
        owner = parallel.owner_table(PART_LENS, world)
        cov, off = _partition_coverage()
        if fail and rank == 1:
            cov = cov.copy(); cov[int(off[2]) + 3] = np.nan          # only rank 1's copy of the coverage is damaged: it fails locally, rank 0 has nothing to complain about
        d = torch.from_numpy(cov).to(cv.device)
        res = {}
        try:
            for undo in (0, 2):
                seg_len, nseg, _ = cv.cbs_sharded(owner, d, off, 0.01, 500, undo=undo, undo_sd=3.0)
                res["cbs%d" % undo] = [seg_len.cpu().numpy()[int(off[c]):int(off[c]) + int(nseg[c])].tolist() for c in range(len(PART_LENS))]
            for germ in (False, True):
                res["wv%d" % int(germ)] = [b.tolist() for b in cv.wavelets_sharded(owner, d, off, is_germline=germ, window=500)]
            res["error"] = None
        except CanvasError as e:
            res["error"] = str(e)
        q.put((rank, owner.tolist(), res))
        dist.destroy_process_group()
    except Exception:                                           # noqa: BLE001
        import traceback
        q.put((rank, "error", traceback.format_exc()))


def _run_partition(fail, list_first=None):
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_partition_worker, args=(r, world, port, q, fail, list_first)) for r in range(world)]
    for p in procs: p.start()
    got = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs: p.join(60)
    for g in got:
        assert g[1] != "error", g[2]
    return got


def test_sharded_cbs_and_wavelets_equal_the_single_rank_result_and_the_oracle():
    import torch
    import oracle_lib as O
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    got = _run_partition(False)
    from canvas_amd import Canvas
    cv = Canvas(0)
    cov, off = _partition_coverage()
    d = torch.from_numpy(cov).to(cv.device)
    nchr = len(PART_LENS)
    assert sorted(set(got[0][1])) == [0, 1]
    for undo in (0, 2):
        seg_len, nseg, _ = cv.cbs(d, off, 0.01, 500, undo=undo, undo_sd=3.0)
        single = [seg_len.cpu().numpy()[int(off[c]):int(off[c]) + int(nseg[c])].tolist() for c in range(nchr)]
        orc = O.cbs_genome([cov[int(off[c]):int(off[c + 1])] for c in range(nchr)], 0.01, 500, threads=4, undo=undo)
        assert [list(map(int, s)) for s in orc[0]] == single
        for rank, _, res in got:
            assert res["error"] is None, res["error"]
            assert res["cbs%d" % undo] == single, (rank, undo)
        assert sum(len(s) for s in single) > nchr                # the planted changes are found
    for germ in (False, True):
        single = [b.tolist() for b in cv.wavelets(d, off, is_germline=germ, window=500)]
        orc = O.wavelets_genome([cov[int(off[c]):int(off[c + 1])] for c in range(nchr)], is_germline=germ, window=500)
        assert [list(map(int, b)) for b in orc] == single
        for rank, _, res in got:
            assert res["wv%d" % int(germ)] == single, (rank, germ)
        assert single[-1] == []                                   # below MinSize: not segmented


@pytest.mark.parametrize("first", [4, 24])
def test_sharded_lists_that_overflow_the_first_exchange_size(first):
    """the list exchange of canvas_cbs_sharded / canvas_wavelets_sharded takes three sizes (8 192 words per rank, 2^18, the hard bound); with a first size of 4 / 24 words the
    lists of this sample overflow once / twice and every rank must step up together and end with the lists of the default sizes"""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ref = _run_partition(False)
    got = _run_partition(False, list_first=first)
    for (r0, _, a), (r1, _, b) in zip(ref, got):
        assert b["error"] is None and a == b, (r0, r1)


def test_sharded_partition_failure_is_collective():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    got = _run_partition(True)                                   # rank 1 holds a NaN: it fails locally, rank 0 must not hang and must fail too
    for rank, _, res in got:
        assert res["error"] is not None, rank
    assert "rank 1 failed" in got[0][2]["error"] or "finite" in got[0][2]["error"]


def test_the_rccl_worker_with_a_one_rank_communicator():
    """the code of test_rccl_all_gather_between_real_ranks (pipeline + sharded CBS + sharded Wavelets over ncclAllGather) on the one GPU every test box has"""
    import torch
    import torch.multiprocessing as mp
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(0, 1, _free_port(), q))
    p.start()
    try:
        g = q.get(timeout=600)
    finally:
        p.join(60)
        if p.is_alive(): p.kill()
    assert g[1] != "error", g[2]
    assert len(g) == 9 and sum(len(s) for s in g[5]) >= len(LENGTHS) and len(g[6]) == len(LENGTHS)
    split, gcw = g[7], g[8]
    assert (split["group_rank"], split["group_size"]) == (0, 1) and split["gathered"] == [0] and split["parent"] == (0, 1)      # canvas_comm_split / canvas_comm_restore over a real RCCL communicator
    from canvas_amd import Canvas
    cv0 = Canvas(0)
    gb, gh, gm, gf = _gcw_inputs(cv0.device)
    g1 = _bins_out(cv0.device)
    _, _, gtot1, gbs1 = cv0.bin_sample_gcweighted(gb, gm, gh, gf, np.array(GCW_LENGTHS, np.int64), GCW_AUTO, 100, -1, out=g1)
    assert (gcw[0], gcw[1]) == (gbs1, gtot1) and all((gcw[2][k].view(np.uint32) == g1[k][:gtot1].cpu().numpy().view(np.uint32)).all() for k in g1)


# ---- the sample axis: one sample of a trio per rank (canvas_allgather_host, canvas_merge_cleaned_sharded)
TRIO_LENS = [1_300_000, 950_000, 600_000]
TRIO_AUTO = np.array([1, 1, 0], np.uint8)


def _trio_inputs(sample):
    """reference (bases, mask) shared by the samples; hits of `sample` (the child, sample 2, carries a deletion)"""
    from canvas_amd import synth
    thr = synth.poisson_thresholds(0.21)
    ref = [synth.generate_chromosome(SEED + 60, c, L, 0.21, thr) for c, L in enumerate(TRIO_LENS)]
    hits = []
    for c, (b, h, m) in enumerate(ref):
        other = synth.generate_chromosome(SEED + 61 + sample, c, TRIO_LENS[c], 0.21, thr)[1]
        poss = np.unpackbits(m.view(np.uint8), bitorder="little")[: TRIO_LENS[c]].astype(bool)
        h2 = np.where(poss, other, 0).astype(np.uint8)
        if c == 0 and sample == 2: h2[400_000:650_000] = h2[400_000:650_000] // 2
        hits.append(h2)
    return [r[0] for r in ref], [r[2] for r in ref], hits


def _trio_worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from canvas_amd import Canvas, parallel, CLEAN_GCNORM, CLEAN_FILTSIZE, CLEAN_OUTLIERS
        cv = Canvas(0)
        parallel.init_host_comm(cv, rank, world)
        bases, masks, hits = _trio_inputs(rank)
        pad = lambda a: np.concatenate([a, np.zeros((-len(a)) % 64, a.dtype)])
        db = [torch.from_numpy(pad(b)).to(cv.device) for b in bases]; dm = [torch.from_numpy(m.view(np.int64).copy()).to(cv.device) for m in masks]
        dh = [torch.from_numpy(pad(h)).to(cv.device) for h in hits]
        r = parallel.pedigree_sample_flow(cv, db, dm, dh, np.array(TRIO_LENS, np.int64), TRIO_AUTO, CLEAN_GCNORM | CLEAN_FILTSIZE | CLEAN_OUTLIERS)
        q.put((rank, dict(bin_size=r["bin_size"], n_binned=r["n_binned"], n_clean=r["n_clean"], n=r["n"], off=[int(x) for x in r["off"]]),
               r["chr"].cpu().numpy(), r["start"].cpu().numpy(), r["stop"].cpu().numpy(), r["count"].cpu().numpy(), r["state"][:r["n"]].cpu().numpy()))
        dist.destroy_process_group()
    except Exception:                                           # noqa: BLE001
        import traceback
        q.put((rank, "error", traceback.format_exc()))


def test_three_samples_on_three_ranks_equal_the_single_gpu_trio_flow():
    import torch
    import torch.multiprocessing as mp
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_trio_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    got = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs: p.join(60)
    for g in got:
        assert g[1] != "error", g[2]
    # the same trio on one GPU with the single-GPU entry points (the flow tests/test_pedigree_flow_gpu.py checks against the oracle hand-off by hand-off)
    from canvas_amd import Canvas, CLEAN_GCNORM, CLEAN_FILTSIZE, CLEAN_OUTLIERS
    cv = Canvas(0)
    lens = np.array(TRIO_LENS, np.int64); nchr = len(TRIO_LENS)
    pad = lambda a: np.concatenate([a, np.zeros((-len(a)) % 64, a.dtype)])
    rates, dev = [], []
    for s in range(3):
        bases, masks, hits = _trio_inputs(s)
        db = [torch.from_numpy(pad(b)).to(cv.device) for b in bases]; dm = [torch.from_numpy(m.view(np.int64).copy()).to(cv.device) for m in masks]
        dh = [torch.from_numpy(pad(h)).to(cv.device) for h in hits]
        _, _, rate = cv.bin_rates(dh, dm, lens)
        rates += [rate[c] for c in range(nchr) if TRIO_AUTO[c]]
        dev.append((db, dm, dh))
    bin_size = cv.bin_size_from_rates(rates, 100)
    cleaned = []
    for db, dm, dh in dev:
        out, per, total = cv.bin_genome(db, dm, dh, lens, bin_size, 3)
        n_clean, _, _ = cv.clean(out, total, TRIO_AUTO, CLEAN_GCNORM | CLEAN_FILTSIZE | CLEAN_OUTLIERS)
        cleaned.append((out, n_clean, int(total)))
    mc, ms, me, mcnt, k = cv.merge_cleaned([o for o, _, _ in cleaned], [n for _, n, _ in cleaned])
    off = cv.chromosome_offsets(mc, k, nchr)
    assert k > 1000 and len({n for _, n, _ in cleaned}) > 1               # the samples lose different bins: the intersection does something
    for rank, info, c_, s_, e_, v_, st_ in got:
        assert info["bin_size"] == bin_size and info["n_binned"] == cleaned[rank][2] and info["n_clean"] == cleaned[rank][1] and info["n"] == k, (rank, info)
        assert info["off"] == [int(x) for x in off]
        assert (c_ == mc[:k].cpu().numpy()).all() and (s_ == ms[:k].cpu().numpy()).all() and (e_ == me[:k].cpu().numpy()).all(), rank
        assert (v_.view(np.uint32) == mcnt[rank][:k].cpu().numpy().view(np.uint32)).all(), rank
        cov = cv.quantize_f2(mcnt[rank], k)
        assert (st_ == cv.hmm_per_sample(cov, off)[:k].cpu().numpy()).all(), rank


def _merge_fail_worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from canvas_amd import Canvas, parallel
        from canvas_amd.lib import CanvasError
        cv = Canvas(0)
        parallel.init_host_comm(cv, rank, world)
        n = 1000
        bins = dict(chr=torch.zeros(n, dtype=torch.int32, device=cv.device), start=torch.arange(n, dtype=torch.int32, device=cv.device) * 100,
                    stop=torch.arange(n, dtype=torch.int32, device=cv.device) * 100 + 100, count=torch.ones(n, dtype=torch.float32, device=cv.device))
        try:
            cv.merge_cleaned_sharded(bins, -7 if rank == 1 else n)          # rank 1 hands in a negative bin count: a local argument error
            q.put((rank, None))
        except CanvasError as e:
            q.put((rank, str(e)))
        dist.destroy_process_group()
    except Exception:                                           # noqa: BLE001
        import traceback
        q.put((rank, "error: " + traceback.format_exc()))


def test_a_failing_sample_fails_the_bin_intersection_on_every_rank():
    import torch
    import torch.multiprocessing as mp
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_merge_fail_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs: p.join(60)
    assert got[0] is not None and "rank 1 failed" in got[0], got
    assert got[1] is not None and "bad arguments" in got[1], got


# ---------------------------------------------------------------- samples x chromosome groups (BASELINE configs[3] on more ranks than samples)
def _inputs_of_sample(device, sample, only=None):
    """sample 0: the inputs above; sample 1: another seed, its deletion on other chromosomes"""
    import torch
    from canvas_amd import synth
    if sample == 0:
        return _inputs(device, only)
    thr = synth.poisson_thresholds(0.21)
    pad = lambda a: np.concatenate([a, np.zeros((-len(a)) % 64, a.dtype)])
    bases, hits, masks = [], [], []
    for c, L in enumerate(LENGTHS):
        if only is not None and c not in only:
            bases.append(None); hits.append(None); masks.append(None); continue
        b, h, m = synth.generate_chromosome(SEED + 77 * sample, c, L, 0.21, thr)
        if c in (1, 3):
            a0, a1 = L // 3, 2 * L // 3
            h = h.copy(); h[a0:a1] = np.where(np.arange(a0, a1) % 2 == 0, h[a0:a1], 0)
        bases.append(torch.from_numpy(pad(b)).to(device)); hits.append(torch.from_numpy(pad(h)).to(device)); masks.append(torch.from_numpy(m.view(np.int64).copy()).to(device))
    return bases, hits, masks


def _grid_worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from canvas_amd import Canvas, parallel
        layout = parallel.sample_groups(world, 2)
        groups = [dist.new_group(ranks=[r for r in range(world) if layout[r][0] == s], backend="gloo") for s in range(2)]
        sample, grank, gsize = layout[rank]
        cv = Canvas(0)
        parallel.init_host_comm(cv, grank, gsize, group=groups[sample])       # the sub-communicator of this sample's ranks
        owner = parallel.owner_table(LENGTHS, gsize)
        mine = [c for c in range(len(LENGTHS)) if owner[c] == grank]
        bases, hits, masks = _inputs_of_sample(cv.device, sample, only=mine)
        out, cov, state, seg = _buffers(cv.device)
        r = cv.sample_pipeline_sharded(owner, bases, masks, hits, np.array(LENGTHS, np.int64), IS_AUTO, out, cov, state, seg, counts_per_bin=100, bin_size=-1, mode=3, flags=FLAGS)
        cv.synchronize()
        n = r["n_out"]
        q.put((rank, sample, (dict(r, off=r["off"].tolist()), {k: v[:n].cpu().numpy() for k, v in out.items()}, cov[:n].cpu().numpy(), state[:n].cpu().numpy(), seg[:n].cpu().numpy())))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:                                           # noqa: BLE001
        import traceback
        q.put((rank, "error", traceback.format_exc()))


def test_two_samples_times_two_chromosome_groups_on_four_ranks():
    """Four ranks, two samples: parallel.sample_groups gives every sample a group of two ranks, each group shards ITS sample's chromosomes and runs the sharded pipeline inside
    its own communicator (host transport: the group's callback; RCCL: canvas_comm_split) at the same time as the other group.  Every rank must hold exactly what one GPU computes
    for its sample."""
    import torch
    import torch.multiprocessing as mp
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grid_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    got = sorted([q.get(timeout=900) for _ in range(world)], key=lambda t: t[0])
    for p in procs: p.join(60)
    for g in got:
        assert g[1] != "error", g[2]
    assert [g[1] for g in got] == [0, 0, 1, 1]
    from canvas_amd import Canvas
    cv = Canvas(0)
    for sample in (0, 1):
        bases, hits, masks = _inputs_of_sample(cv.device, sample)
        out, cov, state, seg = _buffers(cv.device)
        r1 = cv.sample_pipeline(bases, masks, hits, np.array(LENGTHS, np.int64), IS_AUTO, out, cov, state, seg, counts_per_bin=100, bin_size=-1, mode=3, flags=FLAGS)
        cv.synchronize(); n = r1["n_out"]
        ref = ({k: v[:n].cpu().numpy() for k, v in out.items()}, cov[:n].cpu().numpy(), state[:n].cpu().numpy(), seg[:n].cpu().numpy())
        for rank, s, (r, o, cv_, st, sg) in got:
            if s != sample: continue
            assert (r["bin_size"], r["total"], r["n_out"], r["nseg"], r["lsd"]) == (r1["bin_size"], r1["total"], r1["n_out"], r1["nseg"], r1["lsd"]), (rank, sample)
            for key in ("chr", "start", "stop", "gc"):
                assert (o[key] == ref[0][key]).all(), (rank, key)
            assert (o["count"].view(np.uint32) == ref[0]["count"].view(np.uint32)).all()
            assert (cv_ == ref[1]).all() and (st == ref[2]).all() and (sg == ref[3]).all()
        assert r1["nseg"] > len(LENGTHS)
    # the two samples differ (the groups did not see each other's data)
    assert got[0][2][0]["total"] != got[2][2][0]["total"] or not np.array_equal(got[0][2][2], got[2][2][2])


def test_rccl_sub_communicator_of_one_rank():
    """canvas_comm_split / canvas_comm_restore on the RCCL transport (a one-rank communicator split into a one-rank group: the same calls as samples x chromosome groups on a node)"""
    import ctypes as C
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from canvas_amd import Canvas, parallel
    cv = Canvas(0)
    buf = (C.c_ubyte * 128)()
    cv._check(cv.lib.canvas_comm_unique_id(buf))
    cv._check(cv.lib.canvas_comm_init(cv.ctx, 0, 1, buf))
    assert parallel.split_library_comm(cv, color=3, key=0) == (0, 1)
    assert cv.lib.canvas_comm_split(cv.ctx, 0, 0) != 0          # no split inside a split
    ref = _single(cv)
    bases, hits, masks = _inputs(cv.device)
    out, cov, state, seg = _buffers(cv.device)
    r = cv.sample_pipeline_sharded(np.zeros(len(LENGTHS), np.int32), bases, masks, hits, np.array(LENGTHS, np.int64), IS_AUTO, out, cov, state, seg, counts_per_bin=100, bin_size=-1, mode=3, flags=FLAGS)
    cv.synchronize()
    n = r["n_out"]
    assert (r["total"], n, r["nseg"]) == (ref[0]["total"], ref[0]["n_out"], ref[0]["nseg"]) and (seg[:n].cpu().numpy() == ref[4]).all()
    assert parallel.restore_library_comm(cv) == (0, 1)
    r = cv.sample_pipeline_sharded(np.zeros(len(LENGTHS), np.int32), bases, masks, hits, np.array(LENGTHS, np.int64), IS_AUTO, out, cov, state, seg, counts_per_bin=100, bin_size=-1, mode=3, flags=FLAGS)
    cv.synchronize()
    assert r["n_out"] == ref[0]["n_out"]


# ---------------------------------------------------------------- CanvasBin alone, sharded (BASELINE configs[4]: the tumour's GCContentWeighted bins)
GCW_LENGTHS = [700_000, 410_001, 300_000, 150_016]
GCW_AUTO = [1, 1, 1, 0]


def _gcw_inputs(device, only=None):
    import torch
    from canvas_amd import synth
    thr = synth.poisson_thresholds(0.21)
    pad = lambda a: np.concatenate([a, np.zeros((-len(a)) % 64, a.dtype)])
    rng = np.random.RandomState(31)
    bases, hits, masks, frag = [], [], [], []
    for c, L in enumerate(GCW_LENGTHS):
        b, h, m = synth.generate_chromosome(SEED + 9, c, L, 0.21, thr)
        f = np.where(h > 0, np.clip(rng.normal(350 + 10 * c, 60, L), 1, 5000), 0).astype(np.int16)      # (drawn for every chromosome: the ranks generate the same data)
        if only is not None and c not in only:
            bases.append(None); hits.append(None); masks.append(None); frag.append(None); continue
        bases.append(torch.from_numpy(pad(b)).to(device)); hits.append(torch.from_numpy(pad(h)).to(device)); masks.append(torch.from_numpy(m.view(np.int64).copy()).to(device))
        frag.append(torch.from_numpy(pad(f)).to(device))
    return bases, hits, masks, frag


def _bins_out(device):
    import torch
    cap = sum(GCW_LENGTHS) // 40 + 64
    return {k: torch.empty(cap, dtype=(torch.float32 if k == "count" else torch.int32), device=device) for k in ("chr", "start", "stop", "gc", "count")}


def _bins_worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from canvas_amd import Canvas, parallel
        cv = Canvas(0)
        parallel.init_host_comm(cv, rank, world)
        owner = parallel.owner_table(GCW_LENGTHS, world)
        mine = [c for c in range(len(GCW_LENGTHS)) if owner[c] == rank]
        bases, hits, masks, frag = _gcw_inputs(cv.device, only=mine)
        out = _bins_out(cv.device)
        res = []
        for mode, bin_size in ((5, -1), (5, 300), (3, -1)):
            bs, total = cv.bin_sample_sharded(owner, bases, masks, hits, np.array(GCW_LENGTHS, np.int64), GCW_AUTO, out, counts_per_bin=100, bin_size=bin_size, mode=mode, fraglens=frag if mode == 5 else None)
            res.append((bs, total, {k: v[:total].cpu().numpy() for k, v in out.items()}))
        q.put((rank, owner.tolist(), res))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:                                           # noqa: BLE001
        import traceback
        q.put((rank, "error", traceback.format_exc()))


def test_sharded_binning_incl_gc_content_weighted_equals_the_single_gpu_call():
    """canvas_bin_sample_sharded on two ranks: the whole genome's bins on every rank, bit-identical to canvas_bin_sample_gcweighted / canvas_bin_sample on one GPU.  Mode 5 needs
    the genome-wide mean fragment size and read-GC profile: two reductions over the ranks in front of the rate table (a derived bin size and a given one)."""
    import torch
    import torch.multiprocessing as mp
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bins_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    got = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs: p.join(60)
    for g in got:
        assert g[1] != "error", g[2]
    assert sorted(set(got[0][1])) == [0, 1]
    from canvas_amd import Canvas
    cv = Canvas(0)
    bases, hits, masks, frag = _gcw_inputs(cv.device)
    lens = np.array(GCW_LENGTHS, np.int64)
    for k, (mode, bin_size) in enumerate(((5, -1), (5, 300), (3, -1))):
        out = _bins_out(cv.device)
        if mode == 5:
            _, per, total, bs = cv.bin_sample_gcweighted(bases, masks, hits, frag, lens, GCW_AUTO, 100, bin_size, out=out)
        else:
            r = cv.bin_sample(bases, masks, hits, lens, GCW_AUTO, 100, bin_size, mode, out=out)
            total, bs = r[2], r[3]
        ref = {kk: v[:total].cpu().numpy() for kk, v in out.items()}
        for rank, _, res in got:
            gbs, gtotal, o = res[k]
            assert (gbs, gtotal) == (bs, total), (rank, mode, bin_size, gbs, gtotal, bs, total)
            for key in ("chr", "start", "stop", "gc"):
                assert (o[key] == ref[key]).all(), (rank, mode, key)
            assert (o["count"].view(np.uint32) == ref["count"].view(np.uint32)).all(), (rank, mode)
        assert total > 1000


# ---------------------------------------------------------------- BASELINE configs[4], chromosomes sharded: tumour (mode 5) + normal -> ratio -> Clean -> CBS
SOM_LENGTHS = [2_500_000, 1_300_001, 1_000_000, 600_000]
SOM_FLAGS = 1 | 2 | 4


def _somatic_inputs(device, only=None):
    import torch
    from canvas_amd import synth
    pad = lambda a: np.concatenate([a, np.zeros((-len(a)) % 64, a.dtype)])
    seed = 20260927 + 5
    thr_t = synth.poisson_thresholds(0.28, purity=0.7); thr_n = synth.poisson_thresholds(0.14, flat=True)
    up = lambda a: torch.from_numpy(pad(a)).to(device)
    bases, masks, hits_t, fl, hits_n = [], [], [], [], []
    for c, L in enumerate(SOM_LENGTHS):
        if only is not None and c not in only:
            for lst in (bases, masks, hits_t, fl, hits_n): lst.append(None)
            continue
        t = synth.generate_chromosome(seed, c, L, 0.28, thr_t, hit_seed=seed + 1000, with_fraglen=True)
        n = synth.generate_chromosome(seed, c, L, 0.14, thr_n, hit_seed=seed + 2000)
        bases.append(up(t[0])); masks.append(torch.from_numpy(t[2].view(np.int64).copy()).to(device)); hits_t.append(up(t[1])); fl.append(up(t[3])); hits_n.append(up(n[1]))
    return bases, masks, hits_t, fl, hits_n


def _somatic_summary(r):
    h = lambda t: t.cpu().numpy()
    return dict(bin_size=int(r["bin_size"]), n_bins=int(r["n_bins"]), n_ratio=int(r["n_ratio"]), n_clean=int(r["n_clean"]), lsf=float(r["library_size_factor"]), nseg=r["nseg"].tolist(),
                stats=[int(v) for v in r["cbs_stats"][:5]], cov=h(r["cov"]), seg_len=h(r["seg_len"]), start=h(r["cleaned"]["start"]), count=h(r["cleaned"]["count"]).view(np.uint32),
                off=np.asarray(r["chr_offset"]).tolist())


def _somatic_worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from canvas_amd import Canvas, parallel
        cv = Canvas(0)
        parallel.init_host_comm(cv, rank, world)
        owner = parallel.owner_table(SOM_LENGTHS, world)
        mine = [c for c in range(len(SOM_LENGTHS)) if owner[c] == rank]
        bases, masks, hits_t, fl, hits_n = _somatic_inputs(cv.device, only=mine)
        r = cv.tumor_normal_flow(bases, masks, hits_t, fl, hits_n, np.array(SOM_LENGTHS, np.int64), [1, 1, 1, 0], SOM_FLAGS, alpha=0.01, nperm=2000, keep=True, owner=owner)
        q.put((rank, owner.tolist(), _somatic_summary(r)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:                                           # noqa: BLE001
        import traceback
        q.put((rank, "error", traceback.format_exc()))


def test_sharded_tumour_normal_flow_equals_the_single_gpu_flow():
    """BASELINE configs[4] with the chromosomes of the pair sharded over two ranks (Canvas.tumor_normal_flow(owner=...)): tumour bins -m GCContentWeighted and the normal's bins
    through canvas_bin_sample_sharded, ratio + CanvasClean on every rank, CBS through canvas_cbs_sharded — every rank ends with the single-GPU flow's bins, coverage, segments and
    random-number consumption (which tests/test_somatic_flow_gpu.py pins to the chained oracle)."""
    import torch
    import torch.multiprocessing as mp
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_somatic_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    got = sorted([q.get(timeout=900) for _ in range(world)], key=lambda t: t[0])
    for p in procs: p.join(60)
    for g in got:
        assert g[1] != "error", g[2]
    assert sorted(set(got[0][1])) == [0, 1]
    from canvas_amd import Canvas
    cv = Canvas(0)
    bases, masks, hits_t, fl, hits_n = _somatic_inputs(cv.device)
    ref = _somatic_summary(cv.tumor_normal_flow(bases, masks, hits_t, fl, hits_n, np.array(SOM_LENGTHS, np.int64), [1, 1, 1, 0], SOM_FLAGS, alpha=0.01, nperm=2000, keep=True))
    assert ref["n_clean"] > 1000 and sum(ref["nseg"]) >= len(SOM_LENGTHS)
    for rank, _, s in got:
        for k in ("bin_size", "n_bins", "n_ratio", "n_clean", "lsf", "nseg", "off"):
            assert s[k] == ref[k], (rank, k, s[k], ref[k])
        for k in ("cov", "start", "count"):
            assert np.array_equal(s[k], ref[k]), (rank, k)
        n = sum(ref["nseg"])
        for c in range(len(SOM_LENGTHS)):
            a = ref["off"][c]
            assert np.array_equal(s["seg_len"][a:a + ref["nseg"][c]], ref["seg_len"][a:a + ref["nseg"][c]]), (rank, c)
    # the counters of canvas_cbs_sharded are this rank's chromosomes only: together they are the single-GPU call's (same arc searches, permutations, edge-test draws)
    for k in (0, 2, 4):
        assert sum(s["stats"][k] for _, _, s in got) == ref["stats"][k], (k, [s["stats"] for _, _, s in got], ref["stats"])


# ---------------------------------------------------------------- the pedigree flow on samples x chromosome groups
def _trio_grid_worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from canvas_amd import Canvas, parallel, CLEAN_GCNORM, CLEAN_FILTSIZE, CLEAN_OUTLIERS
        layout = parallel.sample_groups(world, 3)                 # four ranks: (0,0,2) (0,1,2) (1,0,1) (2,0,1)
        groups = [dist.new_group(ranks=[r for r in range(world) if layout[r][0] == s], backend="gloo") for s in range(3)]
        sample, grank, gsize = layout[rank]
        cv = Canvas(0)
        enter_world = lambda: parallel.init_host_comm(cv, rank, world)
        enter_group = lambda: parallel.init_host_comm(cv, grank, gsize, group=groups[sample])
        owner = parallel.owner_table(TRIO_LENS, gsize)
        bases, masks, hits = _trio_inputs(sample)
        pad = lambda a: np.concatenate([a, np.zeros((-len(a)) % 64, a.dtype)])
        up = lambda lst, f: [f(x) if owner[c] == grank else None for c, x in enumerate(lst)]
        db = up(bases, lambda b: torch.from_numpy(pad(b)).to(cv.device)); dm = up(masks, lambda m: torch.from_numpy(m.view(np.int64).copy()).to(cv.device))
        dh = up(hits, lambda h: torch.from_numpy(pad(h)).to(cv.device))
        r = parallel.pedigree_grid_flow(cv, layout, rank, enter_world, enter_group, db, dm, dh, np.array(TRIO_LENS, np.int64), TRIO_AUTO, CLEAN_GCNORM | CLEAN_FILTSIZE | CLEAN_OUTLIERS)
        q.put((rank, dict(bin_size=r["bin_size"], n_binned=r["n_binned"], n_clean=r["n_clean"], n=r["n"], off=[int(x) for x in r["off"]], sample=sample),
               r["chr"].cpu().numpy(), r["start"].cpu().numpy(), r["stop"].cpu().numpy(), r["count"].cpu().numpy(), r["state"][:r["n"]].cpu().numpy()))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:                                           # noqa: BLE001
        import traceback
        q.put((rank, "error", traceback.format_exc()))


def test_a_trio_on_four_ranks_as_samples_times_chromosome_groups():
    """BASELINE configs[3] on more ranks than samples (parallel.pedigree_grid_flow): the first sample's chromosomes are sharded over two ranks, the other two samples have a rank
    each; the bin size and the bin intersection span all four ranks, CanvasBin and PerSampleHMM run inside the sample's group (canvas_bin_sample_sharded,
    canvas_hmm_per_sample_sharded).  Every rank must end with what the single-GPU trio flow gives for its sample."""
    import torch
    import torch.multiprocessing as mp
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_trio_grid_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    got = sorted([q.get(timeout=900) for _ in range(world)], key=lambda t: t[0])
    for p in procs: p.join(60)
    for g in got:
        assert g[1] != "error", g[2]
    assert [g[1]["sample"] for g in got] == [0, 0, 1, 2]
    from canvas_amd import Canvas, CLEAN_GCNORM, CLEAN_FILTSIZE, CLEAN_OUTLIERS
    cv = Canvas(0)
    lens = np.array(TRIO_LENS, np.int64); nchr = len(TRIO_LENS)
    pad = lambda a: np.concatenate([a, np.zeros((-len(a)) % 64, a.dtype)])
    rates, dev = [], []
    for s in range(3):
        bases, masks, hits = _trio_inputs(s)
        db = [torch.from_numpy(pad(b)).to(cv.device) for b in bases]; dm = [torch.from_numpy(m.view(np.int64).copy()).to(cv.device) for m in masks]
        dh = [torch.from_numpy(pad(h)).to(cv.device) for h in hits]
        _, _, rate = cv.bin_rates(dh, dm, lens)
        rates += [rate[c] for c in range(nchr) if TRIO_AUTO[c]]
        dev.append((db, dm, dh))
    bin_size = cv.bin_size_from_rates(rates, 100)
    cleaned = []
    for db, dm, dh in dev:
        out, per, total = cv.bin_genome(db, dm, dh, lens, bin_size, 3)
        n_clean, _, _ = cv.clean(out, total, TRIO_AUTO, CLEAN_GCNORM | CLEAN_FILTSIZE | CLEAN_OUTLIERS)
        cleaned.append((out, n_clean, int(total)))
    mc, ms, me, mcnt, k = cv.merge_cleaned([o for o, _, _ in cleaned], [n for _, n, _ in cleaned])
    off = cv.chromosome_offsets(mc, k, nchr)
    for rank, info, c_, s_, e_, v_, st_ in got:
        s = info["sample"]
        assert info["bin_size"] == bin_size and info["n_binned"] == cleaned[s][2] and info["n_clean"] == cleaned[s][1] and info["n"] == k, (rank, info)
        assert info["off"] == [int(x) for x in off]
        assert (c_ == mc[:k].cpu().numpy()).all() and (s_ == ms[:k].cpu().numpy()).all() and (e_ == me[:k].cpu().numpy()).all(), rank
        assert (v_.view(np.uint32) == mcnt[s][:k].cpu().numpy().view(np.uint32)).all(), rank
        cov = cv.quantize_f2(mcnt[s], k)
        assert (st_ == cv.hmm_per_sample(cov, off)[:k].cpu().numpy()).all(), rank


# ---- PerSampleHMM sharded on its own (canvas_hmm_per_sample_sharded): the state runs are recorded, gathered and expanded on the device
HMM_LENS = [9_000, 0, 40_000, 3, 700, 15_000, 12]


def _hmm_coverage():
    rng = np.random.default_rng(SEED + 31)
    parts = []
    for c, L in enumerate(HMM_LENS):
        x = rng.normal(60.0, 5.0, L)
        if L == 40_000:                                          # a state change every fifty bins: 800 runs, more than the exchange's first bound (330 records) holds — every rank retries with the hard bound
            x = np.where((np.arange(L) // 50) % 2 == 0, 60.0, 120.0) + rng.normal(0, 1.0, L)
        parts.append(np.round(np.clip(x, 0, None), 2))
    cov = np.concatenate(parts) if parts else np.zeros(0)
    off = np.concatenate([[0], np.cumsum(HMM_LENS)]).astype(np.int64)
    return cov, off


def _hmm_worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from canvas_amd import Canvas, parallel
        cv = Canvas(0)
        parallel.init_host_comm(cv, rank, world)
        cov, off = _hmm_coverage()
        owner = np.array([c % world for c in range(len(HMM_LENS))], np.int32)
        d = torch.from_numpy(cov).to(cv.device)
        st = cv.hmm_per_sample_sharded(owner, d, off).cpu().numpy()
        single = cv.hmm_per_sample(d, off).cpu().numpy()
        q.put((rank, st.tolist(), bool((st == single).all()), [int(v) for v in cv.sharded_stats()]))
        dist.destroy_process_group()
    except Exception:                                           # noqa: BLE001
        import traceback
        q.put((rank, "error", traceback.format_exc(), None))


def test_sharded_per_sample_hmm_keeps_its_state_runs_on_the_device():
    """canvas_hmm_per_sample_sharded on three ranks (host transport, one GPU): chromosomes of 0, 3 and 12 bins, one whose path changes state every fifty bins (the first bound of
    the record exchange overflows and every rank retries with the hard one); every rank's states equal the single-rank call and the oracle's paths"""
    import torch.multiprocessing as mp
    import oracle_lib as O
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hmm_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    got = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs: p.join(60)
    for g in got:
        assert g[1] != "error", g[2]
    cov, off = _hmm_coverage()
    per = [np.ascontiguousarray(cov[off[c]:off[c + 1]]) for c in range(len(HMM_LENS))]
    paths, ran = O.hmm_genome_per_sample(per, threads=4)
    exp = np.concatenate([paths[c] if ran[c] else np.full(len(per[c]), -1, np.int32) for c in range(len(HMM_LENS))])
    for rank, st, same, stats in got:
        assert same and (np.array(st, np.int32) == exp).all(), rank
        assert stats[0] == world
    assert got[2][3][4] > 500                                    # rank 2 owns the 40 000-bin chromosome: its runs went through the retry


def _hmm_fail_worker(rank, world, port, q, inject):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        os.environ["CANVAS_HMM_SHARDED_FAIL_RESERVE"] = inject
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from canvas_amd import Canvas, parallel
        from canvas_amd.lib import CanvasError
        cv = Canvas(0)
        parallel.init_host_comm(cv, rank, world)
        cov, off = _hmm_coverage()
        owner = np.array([c % world for c in range(len(HMM_LENS))], np.int32)
        d = torch.from_numpy(cov).to(cv.device)
        try:
            cv.hmm_per_sample_sharded(owner, d, off)
            q.put((rank, "returned", ""))
        except CanvasError as e:
            q.put((rank, "failed", str(e)))
        dist.destroy_process_group()
    except Exception:                                           # noqa: BLE001
        import traceback
        q.put((rank, "error", traceback.format_exc()))


@pytest.mark.parametrize("inject", ["1:0", "1:1"])
def test_sharded_hmm_a_failed_reservation_on_one_rank_fails_every_rank(inject):
    """ADVICE r05 (sharded.hip:583): rank 1's workspace reservation fails — on the first attempt, or on the retry with the hard bound (W x 4 (N + nchr) words: where an
    out-of-memory is plausible) — and must be announced THROUGH the collective: every rank returns an error, nobody is left waiting in the all-gather"""
    import torch
    import torch.multiprocessing as mp
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hmm_fail_worker, args=(r, world, port, q, inject)) for r in range(world)]
    for p in procs: p.start()
    try:
        got = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])      # a hang shows up as queue.Empty here
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive(): p.kill()
    for g in got:
        assert g[1] == "failed", g
    assert "injected" in got[1][2] and "rank 1 failed" in got[0][2] and "rank 1 failed" in got[2][2]
