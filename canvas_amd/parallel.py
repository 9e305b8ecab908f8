"""Multi-process plumbing of the hot path (one process per GPU).

Two ways to use N GPUs:
  * chromosome sharding of ONE sample (north_star; BASELINE configs[3], [4]): rank r holds an LPT group of chromosomes, the library's
    canvas_sample_pipeline_sharded does the three exchanges (rate table, bins, segment boundaries) over RCCL; `owner_table` gives the assignment;
  * a cohort: one sample per rank, no data-path collective.
The bookkeeping below works on any torch.distributed backend (nccl on the GPU box, gloo in the CPU tests)."""
import ctypes as C

import numpy as np


def shard_units(weights, world):
    """Longest-processing-time assignment of units (e.g. chromosomes by length) to ranks: returns a list of index lists."""
    order = np.argsort(-np.asarray(weights, dtype=np.float64), kind="stable")
    load = np.zeros(world)
    out = [[] for _ in range(world)]
    for i in order:
        r = int(np.argmin(load))
        out[r].append(int(i))
        load[r] += weights[i]
    return [sorted(o) for o in out]


def owner_table(lengths, world):
    """owner[c] = rank that holds chromosome c (LPT on length): the h_chr_owner argument of canvas_sample_pipeline_sharded"""
    owner = np.zeros(len(lengths), np.int32)
    for r, units in enumerate(shard_units(lengths, world)):
        for c in units:
            owner[c] = r
    return owner


def sample_groups(world, nsamples, weights=None):
    """Samples x chromosome groups (BASELINE configs[3]: a trio on 8 GPUs): the ranks are dealt to the samples in contiguous groups whose sizes follow the samples' weights
    (equal by default), every sample gets at least one rank and no rank stays idle while world >= nsamples.  Returns one (sample, rank in the group, group size) per world
    rank; with world < nsamples the samples are dealt round-robin instead and every group has one rank (several samples per rank: the caller loops)."""
    if world <= nsamples:
        return [(r, 0, 1) for r in range(world)]
    w = np.ones(nsamples) if weights is None else np.asarray(weights, dtype=np.float64)
    sizes = np.ones(nsamples, dtype=np.int64)
    for _ in range(world - nsamples):                       # the next rank goes to the sample with the most work per rank
        sizes[int(np.argmax(w / sizes))] += 1
    out = []
    for s in range(nsamples):
        out += [(s, k, int(sizes[s])) for k in range(int(sizes[s]))]
    return out


def split_library_comm(cv, color, key):
    """canvas_comm_split: the RCCL communicator of init_library_comm split by color (ranks of one sample), ordered by key; sharded calls run inside it until restore_library_comm"""
    cv._check(cv.lib.canvas_comm_split(cv.ctx, int(color), int(key)))
    r, n = C.c_int32(0), C.c_int32(0)
    cv._check(cv.lib.canvas_comm_rank(cv.ctx, C.byref(r), C.byref(n)))
    cv.comm_size = n.value
    return r.value, n.value


def restore_library_comm(cv):
    cv._check(cv.lib.canvas_comm_restore(cv.ctx))
    r, n = C.c_int32(0), C.c_int32(0)
    cv._check(cv.lib.canvas_comm_rank(cv.ctx, C.byref(r), C.byref(n)))
    cv.comm_size = n.value
    return r.value, n.value


def sample_seed(base_seed, rank):
    """cohort mode: rank r owns sample r"""
    return base_seed + 1000 * rank


def _gloo():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_backend() == "gloo"


def all_gather_tensor(t):
    """dist.all_gather of one small tensor on either backend (gloo has no all_gather of device tensors: they travel through host memory); list of tensors on t's device"""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return [t]
    src = t.cpu() if _gloo() else t
    outs = [torch.zeros_like(src) for _ in range(world)]
    dist.all_gather(outs, src)
    return [o.to(t.device) for o in outs]


def aggregate_throughput(seconds, units, device=None):
    """(max over ranks of the timed region, sum over ranks of the units processed) -> (seconds, units, units/s)"""
    import torch
    import torch.distributed as dist
    if _gloo(): device = None
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    u = torch.tensor([float(units)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item()), float(u.item()) / float(t.item())


def max_over_ranks(seconds, device=None):
    import torch
    import torch.distributed as dist
    if _gloo(): device = None
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_boundary_records(local_records, max_per_rank):
    """reference semantics of canvas_allgather_boundaries on any backend: every rank contributes [count, rec...] padded to
    1 + max_per_rank int32; returns (counts per rank, [records of rank r])"""
    import torch
    import torch.distributed as dist
    rec = torch.zeros(1 + max_per_rank, dtype=torch.int32)
    rec[0] = len(local_records)
    rec[1:1 + len(local_records)] = torch.as_tensor(local_records, dtype=torch.int32)
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return [len(local_records)], [list(local_records)]
    out = [torch.zeros_like(rec) for _ in range(world)]
    dist.all_gather(out, rec)
    counts = [int(o[0]) for o in out]
    return counts, [o[1:1 + c].tolist() for o, c in zip(out, counts)]


# ---- the bookkeeping of canvas_sample_pipeline_sharded, restated on host arrays (what the library does between its kernels; exercised on gloo by the CPU tests)
def bin_layout(owner, pop, pop_before, bin_size, world):
    """from the exchanged rate table: bins per chromosome, file-order offsets, offset of every chromosome inside its owner's packed block, bins per rank"""
    nb = [(int(p) - int(b)) // int(bin_size) for p, b in zip(pop, pop_before)]
    bin_off = np.concatenate([[0], np.cumsum(nb)]).astype(np.int64)
    rank_off = np.zeros(len(owner), np.int64); per_rank = np.zeros(world, np.int64)
    for c, r in enumerate(owner):
        rank_off[c] = per_rank[r]; per_rank[r] += nb[c]
    return np.array(nb, np.int64), bin_off, rank_off, per_rank


def boundary_records(chroms, states_per_chrom):
    """[(chr, startBin, endBin, state) ...] of the chromosomes a rank owns: one record per run of equal states, flat int32 list"""
    out = []
    for c, st in zip(chroms, states_per_chrom):
        st = np.asarray(st)
        if len(st) == 0:
            continue
        starts = np.concatenate([[0], np.nonzero(st[1:] != st[:-1])[0] + 1])
        ends = np.concatenate([starts[1:] - 1, [len(st) - 1]])
        for a, b in zip(starts, ends):
            out += [int(c), int(a), int(b), int(st[a])]
    return out


def states_from_records(records_per_rank, owner, chr_offset):
    """every rank rebuilds the state of every bin of the genome from the gathered records"""
    n = int(chr_offset[-1])
    state = np.full(n, -9, np.int32)
    for recs in records_per_rank:
        for k in range(0, len(recs), 4):
            c, a, b, s = recs[k:k + 4]
            state[chr_offset[c] + a:chr_offset[c] + b + 1] = s
    return state


def segment_ids_from_states(state, chr_offset, start, stop, max_inter_bin_dist=1000000):
    """SegmentationResultsProcessor.PostProcessSegments without forbidden intervals / ploidy: the running id in file order (Q17)"""
    seg = np.zeros(len(state), np.int32); cur = -1
    for c in range(len(chr_offset) - 1):
        prev_end = 0
        for i in range(int(chr_offset[c]), int(chr_offset[c + 1])):
            first = i == chr_offset[c]
            new = state[i] >= 0 and (first or state[i - 1] != state[i])
            if not new and not first and prev_end > 0 and max_inter_bin_dist >= 0 and prev_end + max_inter_bin_dist < int(start[i]):
                new = True
            if new:
                cur += 1
            seg[i] = cur; prev_end = int(stop[i])
    return seg


# ---- communicators of the library
def init_library_comm(cv, rank, world):
    """RCCL communicator of the library's own collectives: rank 0 creates the unique id, torch.distributed carries the 128 bytes"""
    import torch.distributed as dist
    ident = [None]
    if rank == 0:
        buf = (C.c_ubyte * 128)()
        cv._check(cv.lib.canvas_comm_unique_id(buf))
        ident[0] = bytes(buf)
    dist.broadcast_object_list(ident, src=0)
    idbuf = (C.c_ubyte * 128).from_buffer_copy(ident[0])
    cv._check(cv.lib.canvas_comm_init(cv.ctx, rank, world, idbuf))
    cv.comm_size = world


_HOST_ALLGATHER = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


def init_host_comm(cv, rank, world, group=None):
    """host-callback transport (canvas_comm_init_host): the exchanges go through torch.distributed on CPU tensors (gloo).  For ranks that cannot form an RCCL
    communicator, e.g. two processes that share one GPU in the tests."""
    import torch
    import torch.distributed as dist

    def cb(user, send, nbytes, recv):
        try:
            s = torch.frombuffer((C.c_ubyte * nbytes).from_address(send), dtype=torch.uint8).clone()
            if world == 1:
                outs = [s]
            else:
                outs = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(outs, s, group=group)
            r = torch.cat(outs).contiguous()
            C.memmove(recv, r.data_ptr(), nbytes * world)
            return 0
        except Exception:                                # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            return 1

    fn = _HOST_ALLGATHER(cb)
    cv._host_allgather_cb = fn                            # keep the trampoline alive as long as the context
    cv.lib.canvas_comm_init_host.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _HOST_ALLGATHER, C.c_void_p]
    cv._check(cv.lib.canvas_comm_init_host(cv.ctx, rank, world, fn, None))
    cv.comm_size = world


# ---- the sample axis: one sample of a pedigree per rank (BASELINE configs[3])
def pedigree_sample_flow(cv, bases, masks, hits, lens, is_autosome, flags, counts_per_bin=100, is_y=None):
    """This rank's sample of a pedigree through CanvasBin -> CanvasClean -> bin intersection -> F2 -> PerSampleHMM; every rank (= sample) calls it.  The samples meet twice:
    canvas_allgather_host (every sample's autosomal rates -> ONE bin size, CanvasBin.cs:86-110) and canvas_merge_cleaned_sharded (the bins every sample still has,
    Utilities.cs:834-920).  Returns dict(bin_size, n_binned, n_clean, chr, start, stop, count (this sample, merged bins), n, off, cov, state)."""
    nchr = len(lens)
    ia = np.ascontiguousarray(is_autosome, np.uint8)
    _, _, rate = cv.bin_rates(hits, masks, lens)
    mine = np.array([rate[c] if ia[c] else -1.0 for c in range(nchr)], np.float64)       # -1: not an autosome (dropped below, like the reference's filter)
    allr = cv.allgather_host(mine)
    rates = [float(allr[r, c]) for r in range(allr.shape[0]) for c in range(nchr) if ia[c]]      # sample by sample, chromosomes in file order: the order of the reference's list
    bin_size = cv.bin_size_from_rates(rates, counts_per_bin)
    out, per, total = cv.bin_genome(bases, masks, hits, lens, bin_size, 3)
    n_clean, _, _ = cv.clean(out, total, is_autosome, flags, is_y=is_y)
    mc, ms, me, mv, k = cv.merge_cleaned_sharded(out, n_clean)
    off = cv.chromosome_offsets(mc, k, nchr)
    cov = cv.quantize_f2(mv, k)
    state = cv.hmm_per_sample(cov, off)
    return dict(bin_size=bin_size, n_binned=int(total), n_clean=int(n_clean), chr=mc, start=ms, stop=me, count=mv, n=int(k), off=off, cov=cov, state=state)


def pedigree_grid_flow(cv, layout, world_rank, enter_world, enter_group, bases, masks, hits, lens, is_autosome, flags, counts_per_bin=100, is_y=None):
    """Samples x chromosome groups (BASELINE configs[3] on more ranks than samples; layout = sample_groups(world, nsamples)): this rank is rank `g` of the group of sample `s`
    and holds the arrays of ITS chromosomes of THAT sample (None elsewhere; owner = owner_table(lens, group size)).  The pedigree's couplings span the world communicator —
    the multi-sample bin size (CanvasBin.cs:86-110: every rank contributes the rates of the chromosomes it owns) and the bin intersection (Utilities.cs:834-920:
    canvas_merge_cleaned_sharded, where the ranks of a group all hold their sample's cleaned bins: a sample that appears twice does not change an intersection) —, CanvasBin
    and PerSampleHMM run sharded inside the sample's group (canvas_bin_sample_sharded, canvas_hmm_per_sample_sharded), CanvasClean redundantly on the group's ranks.
    enter_world() / enter_group() switch the context's communicator (RCCL: restore_library_comm / split_library_comm; host transport: init_host_comm with the group).
    Returns the dict of pedigree_sample_flow; every rank of a group ends with its sample's result."""
    import torch
    sample, grank, gsize = layout[world_rank]
    world = len(layout)
    nchr = len(lens)
    ia = np.ascontiguousarray(is_autosome, np.uint8)
    owner = owner_table(lens, gsize)
    mine = [c for c in range(nchr) if owner[c] == grank]
    # ---- one bin size for the pedigree: the autosomal rates of every sample, each chromosome from the rank that owns it
    enter_world()
    rates_mine = np.full(nchr, -2.0, np.float64)                    # -2: not mine
    if mine:
        _, _, r = cv.bin_rates([hits[c] for c in mine], [masks[c] for c in mine], [lens[c] for c in mine])
        for i, c in enumerate(mine):
            rates_mine[c] = r[i] if ia[c] else -1.0                 # -1: not an autosome
    allr = cv.allgather_host(rates_mine)
    rates = []
    for s in sorted(set(l[0] for l in layout)):                      # sample by sample, chromosomes in file order: the order of the reference's list
        ranks = [r for r in range(world) if layout[r][0] == s]
        ow = owner_table(lens, len(ranks))
        for c in range(nchr):
            if ia[c]:
                rates.append(float(allr[ranks[int(ow[c])], c]))
    bin_size = cv.bin_size_from_rates(rates, counts_per_bin)
    # ---- CanvasBin of the sample inside its group, CanvasClean on every rank of the group
    enter_group()
    cap = int(sum(int(L) for L in lens) // max(1, bin_size)) + nchr + 64
    mk = lambda dt: torch.empty(cap, dtype=dt, device=cv.device)
    out = dict(chr=mk(torch.int32), start=mk(torch.int32), stop=mk(torch.int32), gc=mk(torch.int32), count=mk(torch.float32))
    _, total = cv.bin_sample_sharded(owner, bases, masks, hits, lens, is_autosome, out, counts_per_bin, bin_size, 3)
    n_clean, _, _ = cv.clean(out, total, is_autosome, flags, is_y=is_y)
    # ---- the bins every sample still has: all ranks
    enter_world()
    mc, ms, me, mv, k = cv.merge_cleaned_sharded(out, n_clean)
    off = cv.chromosome_offsets(mc, k, nchr)
    cov = cv.quantize_f2(mv, k)
    # ---- PerSampleHMM inside the group
    enter_group()
    state = cv.hmm_per_sample_sharded(owner, cov, off)
    return dict(bin_size=bin_size, n_binned=int(total), n_clean=int(n_clean), chr=mc, start=ms, stop=me, count=mv, n=int(k), off=off, cov=cov, state=state, sample=sample)


# ---- bench.py --gpus N --multi sharded
def bench_sharded(args, cv, rank, world, device, hbm_peak_gbs):
    """ONE 60x sample sharded by chromosome over the ranks (strong scaling); afterwards rank 0 runs the same sample on its own GPU and compares every output,
    and every rank runs the cohort mode (one sample per rank) so that both scalings come out of one launch."""
    import json
    import time
    import torch
    import torch.distributed as dist
    from . import synth, CLEAN_GCNORM, CLEAN_FILTSIZE, CLEAN_OUTLIERS, CLEAN_LOCALSD
    from .lib import synth_generate_device

    seed = 20260927 + 3
    lengths = [max(200_000, int(L * args.scale)) for L in synth.GRCH38]
    nchr = len(lengths); lens = np.array(lengths, np.int64); total_bases = int(lens.sum())
    owner = owner_table(lengths, world)
    is_auto = synth.IS_AUTOSOME
    flags = CLEAN_GCNORM | CLEAN_FILTSIZE | CLEAN_OUTLIERS | CLEAN_LOCALSD
    thr = None
    bases, hits, masks = [None] * nchr, [None] * nchr, [None] * nchr
    for c in range(nchr):
        if owner[c] == rank:
            bases[c], hits[c], masks[c], thr = synth_generate_device(seed, c, lengths[c], args.rate, device, thr)
    torch.cuda.synchronize()
    cap = int(total_bases // 100) + 16
    mk = lambda dt: torch.empty(cap, dtype=dt, device=device)
    out = dict(chr=mk(torch.int32), start=mk(torch.int32), stop=mk(torch.int32), gc=mk(torch.int32), count=mk(torch.float32))
    cov, state, seg = mk(torch.float64), mk(torch.int32), mk(torch.int32)

    def barrier():
        cv.synchronize(); torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()

    # ---- cohort mode FIRST: one sample per rank (rank r: the cohort's sample r), no library collective.  Rank 0's cohort sample IS the sample that is sharded below,
    # so its single-GPU result is the reference the sharded result is compared with, and this mode's line is what gets printed should the sharded mode fail.
    cseed = sample_seed(seed, rank)
    cb, ch, cm = [], [], []
    for c in range(nchr):
        if cseed == seed and bases[c] is not None:
            cb.append(bases[c]); ch.append(hits[c]); cm.append(masks[c])
        else:
            b_, h_, m_, thr = synth_generate_device(cseed, c, lengths[c], args.rate, device, thr)
            cb.append(b_); ch.append(h_); cm.append(m_)
    torch.cuda.synchronize()
    prepared = None

    def cstep():
        nonlocal prepared
        rr = cv.sample_pipeline(cb, cm, ch, lens, is_auto, out, cov, state, seg, counts_per_bin=100, bin_size=-1, mode=3, flags=flags, prepared=prepared)
        prepared = rr["prepared"]
        cv.synchronize()
        return rr

    for _ in range(args.warmup):
        cstep()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rr = cstep()
    barrier()
    cdt, cbins, _ = aggregate_throughput(time.perf_counter() - t0, float(rr["total"]), device=device)
    n1 = int(rr["n_out"])
    single = dict(n=n1, nseg=int(rr["nseg"]), total=int(rr["total"]), seg=seg[:n1].clone(), state=state[:n1].clone(), count=out["count"][:n1].clone(), start=out["start"][:n1].clone()) if rank == 0 else None
    cohort = {"value": round(cbins / (cdt / args.steps), 1), "ms_per_step": round(cdt / args.steps * 1e3, 3), "scaling": "weak", "samples": world,
              "note": "one 60x sample per rank, no data-path collective (python bench.py --multi cohort prints this mode as the headline)"}
    base = {"metric": "genome-bins/sec (bin+clean+partition)", "unit": "bins/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "vs_baseline": None,
            "dtype": "u8/int32 (bin), f32/f64 (clean, viterbi)", "data": "synthetic",
            "transport": "host callback over gloo, every rank on GPU 0 (CANVAS_BENCH_ONE_GPU: the tests' launch, not a scaling measurement)" if _gloo() else "RCCL (ncclAllGather), one GPU per rank"}

    def fallback_line(why):
        """the sharded mode did not complete: the cohort mode (already measured) becomes the line of this launch, with the reason"""
        if rank == 0:
            print(json.dumps({**base, "value": cohort["value"], "ms_per_step": cohort["ms_per_step"], "scaling": "weak",
                              "config": {"workload": "BASELINE configs[2]: whole-genome GRCh38 60x, one sample per rank (cohort mode)", "bases_per_sample": total_bases, "samples": world,
                                         "scale": args.scale, "rate": args.rate, "multi": "cohort"},
                              "sharded_error": why, "cohort_mode": cohort}), flush=True)

    # a collective that never returns must not cost the launch its line: after CANVAS_SHARDED_TIMEOUT seconds (default 240) every rank leaves, rank 0 prints the cohort line first
    import os
    import sys
    import threading
    done = threading.Event()

    def watchdog():
        if not done.wait(float(os.environ.get("CANVAS_SHARDED_TIMEOUT", "240"))):
            fallback_line("the sharded pipeline did not finish within the watchdog's limit")
            os._exit(3)                                           # the line carries the cohort number and the reason; the exit status says the sharded mode failed

    threading.Thread(target=watchdog, daemon=True).start()
    try:
        def step():
            r = cv.sample_pipeline_sharded(owner, bases, masks, hits, lens, is_auto, out, cov, state, seg, counts_per_bin=100, bin_size=-1, mode=3, flags=flags)
            cv.synchronize()
            return r

        for _ in range(args.warmup):
            step()
        cv.profile_enable(True)
        for name in ("bin_summary", "allgather", "clean_total"):
            cv.profile_get(name, reset=True)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            r = step()
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0, device)
        ms_sum, k_sum = cv.profile_get("bin_summary"); ms_ag, k_ag = cv.profile_get("allgather"); ms_cl, k_cl = cv.profile_get("clean_total")
        st = cv.sharded_stats()
        sharded = {"seconds_per_pass": dt / args.steps, "bins": int(r["total"]), "n_out": int(r["n_out"]), "nseg": int(r["nseg"]), "bin_size": int(r["bin_size"])}
        # every rank must hold the same result: compare a digest of the segment ids and the cleaned counts across ranks
        n = int(r["n_out"])
        dig = torch.stack([seg[:n].to(torch.int64).sum(), (seg[:n].to(torch.int64) * torch.arange(n, device=device) % 1000003).sum(), state[:n].to(torch.int64).sum(),
                           out["count"][:n].view(torch.int32).to(torch.int64).sum()])
        digs = all_gather_tensor(dig)
        same_on_all_ranks = bool(all((d == digs[0]).all() for d in digs))
        equals_single = None
        if rank == 0:
            equals_single = bool(single["n"] == n and single["nseg"] == sharded["nseg"] and single["total"] == sharded["bins"] and (seg[:n] == single["seg"]).all()
                                 and (state[:n] == single["state"]).all() and (out["count"][:n].view(torch.int32) == single["count"].view(torch.int32)).all()
                                 and (out["start"][:n] == single["start"]).all())
    except Exception as e:                                        # a library error on this rank: the others are stopped by their watchdogs
        done.set()
        fallback_line("%s: %s" % (type(e).__name__, e))
        os._exit(3)
    done.set()
    wrong = [None]
    if rank == 0 and not (same_on_all_ranks and equals_single):
        wrong[0] = "the sharded result is %s" % ("not the same on every rank" if not same_on_all_ranks else "not the single-GPU result")
    dist.broadcast_object_list(wrong, src=0)
    if wrong[0]:
        fallback_line(wrong[0])
        os._exit(4)
    if rank == 0:
        value = sharded["bins"] / sharded["seconds_per_pass"]
        # the dominant kernel on rank 0: the single sweep over the per-base arrays of the chromosomes it owns (2.125 B/base + 4 B per 64 positions + 16 B per 4096-position tile)
        own_bases = int(sum(lengths[c] for c in range(nchr) if owner[c] == 0))
        tiles = int(sum((lengths[c] + 4095) // 4096 for c in range(nchr) if owner[c] == 0))
        alg = 2.125 * own_bases + 4.0 * (own_bases / 64.0) + 16.0 * tiles
        avg_ms = ms_sum / max(1, k_sum)
        rank0_roofline = {"kernel": "k_tile_summary (rank 0: its own chromosomes)", "bound": "hbm", "achieved": round(alg / max(1e-9, avg_ms * 1e-3) / 1e9, 1), "peak": hbm_peak_gbs, "unit": "GB/s",
                          "frac": round(alg / max(1e-9, avg_ms * 1e-3) / 1e9 / hbm_peak_gbs, 4), "traffic": None, "avg_ms": round(avg_ms, 4), "launches": int(k_sum), "algorithmic_bytes": alg}
        result = {**base, "value": round(value, 1), "ms_per_step": round(sharded["seconds_per_pass"] * 1e3, 3), "scaling": "strong",
                  "config": {"workload": "BASELINE configs[2] sharded as configs[3]/[4] prescribe: ONE whole-genome GRCh38 60x sample, chromosomes LPT-sharded over the ranks, "
                                         "rate-table + bins + segment-boundary all-gathers over RCCL",
                             "bases_per_sample": total_bases, "bins_per_sample": sharded["bins"], "bins_after_clean": sharded["n_out"], "bin_size": sharded["bin_size"],
                             "partition": "PerSampleHMM", "clean_flags": "-g -s -r --local-sd-metric-file", "segments": sharded["nseg"], "multi": "sharded",
                             "owner_of_chromosome": [int(x) for x in owner], "scale": args.scale, "rate": args.rate},
                  "roofline": rank0_roofline,
                  "sharded": {"collectives_per_pass": int(round(k_ag / max(1, args.steps))), "collectives_ms_per_pass_rank0": round(ms_ag / max(1, args.steps), 4),
                              "clean_ms_per_pass_rank0": round(ms_cl / max(1, k_cl), 4), "collective_ms": round(ms_ag / max(1, args.steps), 4), "redundant_clean_ms": round(ms_cl / max(1, k_cl), 4),
                              "explains": "a pass cannot drop below redundant_clean_ms (CanvasClean's order statistics are genome-wide: every rank runs it on all bins) + collective_ms (rate table, bins, segment boundaries); "
                                          "what shards is the sweep (k_tile_summary over the owned chromosomes) and the Viterbi passes", "chromosomes_owned_rank0": int(st[1]), "bins_binned_rank0": int(st[2]), "bins_allgather_bytes_per_rank": int(st[3]), "boundary_records_rank0": int(st[4]),
                              "boundary_allgather_bytes_per_rank": int(st[5]), "identical_on_all_ranks": same_on_all_ranks, "equals_single_gpu_result": equals_single,
                              "note": "CanvasClean runs redundantly on every rank (its order statistics are genome-wide): the pass cannot drop below Clean + the collectives"},
                  "cohort_mode": cohort}
    # ---- CanvasPartition -m CBS / -m Wavelets sharded the same way, on the coverage the pipeline left on every rank (untimed for `value`; one warm call, one timed call each).
    # A hang here must not cost the launch its line: after 300 s rank 0 prints the line without this leg and every rank leaves.
    result = result if rank == 0 else None
    leg_done = threading.Event()

    def leg_watchdog():
        if not leg_done.wait(float(os.environ.get("CANVAS_SHARDED_PARTITION_TIMEOUT", "420"))):
            if rank == 0:
                result["partition_sharded"] = {"error": "did not finish within the watchdog's limit"}
                print(json.dumps(result), flush=True)
            os._exit(5)            # the line is out, but a hung sharded CBS / Wavelets / pedigree leg is a failure of the launch (as a hung pipeline leg is: 3 / 4)

    threading.Thread(target=leg_watchdog, daemon=True).start()
    part = {}
    try:
        n = int(r["n_out"]); off = r["off"]
        for name, call, single_call in (("cbs", lambda: cv.cbs_sharded(owner, cov, off, 0.01, 10000), lambda: cv.cbs(cov, off, 0.01, 10000)),
                                        ("wavelets", lambda: cv.wavelets_sharded(owner, cov, off), lambda: cv.wavelets(cov, off))):
            call(); barrier()
            cv.profile_get("allgather", reset=True)
            t0 = time.perf_counter(); got = call(); barrier()
            sec = max_over_ranks(time.perf_counter() - t0, device)
            ms_coll, k_coll = cv.profile_get("allgather")      # hipEvents around every collective of the call, on the library's stream (rank 0's view)
            if name == "cbs":
                flat = got[0][:n].to(torch.int64); dig = torch.stack([flat.sum(), (flat * (torch.arange(n, device=device) % 1009)).sum(), torch.tensor(int(got[1].sum()), device=device)])
            else:
                allbp = np.concatenate([np.asarray(b, np.int64) for b in got] + [np.zeros(1, np.int64)])
                dig = torch.tensor([int(allbp.sum()), int((allbp * (np.arange(len(allbp)) % 1009)).sum()), len(allbp)], device=device)
            digs = all_gather_tensor(dig)
            same = bool(all((d == digs[0]).all() for d in digs))
            eq = None; sec1 = None
            if rank == 0:
                single_call(); t1 = time.perf_counter(); one = single_call(); sec1 = time.perf_counter() - t1
                if name == "cbs":
                    eq = bool((one[0][:n] == got[0][:n]).all() and (one[1] == got[1]).all())
                else:
                    eq = bool(len(one) == len(got) and all(np.array_equal(a, b) for a, b in zip(one, got)))
            part[name] = {"seconds": round(sec, 4), "single_gpu_seconds_rank0": None if sec1 is None else round(sec1, 4), "collectives": int(k_coll), "collective_ms": round(ms_coll, 4),
                          "identical_on_all_ranks": same, "equals_single_gpu_result": eq}
            barrier()
    except Exception as e:                                        # noqa: BLE001
        part["error"] = "%s: %s" % (type(e).__name__, e)
    # ---- the sample axis (BASELINE configs[3]): a pedigree of `world` members over ONE reference, one sample per rank — CanvasBin / CanvasClean / PerSampleHMM local, the
    # multi-sample bin size and the bin intersection as exchanges (canvas_allgather_host, canvas_merge_cleaned_sharded); rank 0 repeats the flow for all members on its own
    # GPU and compares.  Untimed for `value`, weak scaling (one 60x sample per rank).
    ped = {}
    try:
        if world <= 16 and "error" not in part:
            from .lib import synth_generate_sample_device
            thr_t = torch.from_numpy(synth.poisson_thresholds(args.rate).view(np.int32)).to(device)
            if rank == 0:
                rb, rm = cb, cm                                   # rank 0's cohort sample was generated from `seed`: its bases / masks ARE the reference
            else:
                rb, rm = [], []
                for c in range(nchr):
                    b_, _, m_, thr = synth_generate_device(seed, c, lengths[c], args.rate, device, thr)
                    rb.append(b_); rm.append(m_)
            sample_hits = lambda s: [synth_generate_sample_device(seed, seed + 3000 + 17 * s, c, int(L), thr_t, device)[0] for c, L in enumerate(lengths)]
            my_hits = sample_hits(rank)
            torch.cuda.synchronize()
            pedigree_sample_flow(cv, rb, rm, my_hits, lens, is_auto, flags)                 # warm
            barrier()
            cv.profile_get("allgather", reset=True)
            t0 = time.perf_counter()
            pr = pedigree_sample_flow(cv, rb, rm, my_hits, lens, is_auto, flags)
            barrier()
            psec = max_over_ranks(time.perf_counter() - t0, device)
            ms_coll_p, k_coll_p = cv.profile_get("allgather")
            k = pr["n"]
            dig = torch.stack([pr["start"].to(torch.int64).sum(), pr["stop"].to(torch.int64).sum(), (pr["chr"].to(torch.int64) * (torch.arange(k, device=device) % 1009)).sum(),
                               torch.tensor(k, device=device), torch.tensor(int(pr["bin_size"]), device=device)])
            digs = all_gather_tensor(dig)
            same = bool(all((d == digs[0]).all() for d in digs))
            binned = all_gather_tensor(torch.tensor([pr["n_binned"]], dtype=torch.int64, device=device))
            eq = None; sec1 = None
            if rank == 0:
                t1 = time.perf_counter()
                allh = [my_hits] + [sample_hits(s_) for s_ in range(1, world)]
                torch.cuda.synchronize()          # the generator runs on torch's stream, the library on its own: the arrays must exist before the library reads them
                rates = []
                for s_ in range(world):
                    _, _, r_ = cv.bin_rates(allh[s_], rm, lens)
                    rates += [r_[c] for c in range(nchr) if is_auto[c]]
                bs1 = cv.bin_size_from_rates(rates, 100)
                outs1, tot1 = [], []
                for s_ in range(world):
                    o_ = dict(chr=mk(torch.int32), start=mk(torch.int32), stop=mk(torch.int32), gc=mk(torch.int32), count=mk(torch.float32))
                    _, _, t_ = cv.bin_genome(rb, rm, allh[s_], lens, bs1, 3, out=o_)
                    outs1.append(o_); tot1.append(int(t_))
                nout1, _, _ = cv.clean_batch(outs1, tot1, is_auto, flags)
                mc1, ms1, me1, mcnt1, k1 = cv.merge_cleaned(outs1, [int(x) for x in nout1])
                sec1 = time.perf_counter() - t1
                checks = {"bin_size": bs1 == pr["bin_size"], "bins_per_sample_rank0": tot1[0] == pr["n_binned"], "bins_after_clean_rank0": int(nout1[0]) == pr["n_clean"], "merged_count": k1 == k}
                if k1 == k:
                    checks.update(start=bool((ms1[:k1] == pr["start"][:k]).all()), stop=bool((me1[:k1] == pr["stop"][:k]).all()), chr=bool((mc1[:k1] == pr["chr"][:k]).all()),
                                  count_bits=bool((mcnt1[0][:k1].view(torch.int32) == pr["count"][:k].view(torch.int32)).all()))
                eq = bool(all(checks.values()))
                ped_mismatch = [name for name, ok in checks.items() if not ok] + ([] if k1 == k else ["merged %d vs %d" % (k1, k)])
                del allh, outs1
            ped = {"samples": world, "seconds": round(psec, 4), "collectives": int(k_coll_p), "collective_ms": round(ms_coll_p, 4), "bins_per_s": round(float(sum(int(b.item()) for b in binned)) / psec, 1), "scaling": "weak", "bin_size": int(pr["bin_size"]),
                   "bins_common_to_all": int(k), "identical_on_all_ranks": same, "equals_single_gpu_flow_rank0": eq, "differs_in": (ped_mismatch if rank == 0 and not eq else None), "single_gpu_seconds_incl_generating_the_other_samples_rank0": None if sec1 is None else round(sec1, 3),
                   "note": "one sample per rank over one reference: rates all-gather -> one bin size, CanvasBin + CanvasClean local, canvas_merge_cleaned_sharded (12 B per bin per rank), F2 + PerSampleHMM local"}
            barrier()
    except Exception as e:                                        # noqa: BLE001
        ped = {"error": "%s: %s" % (type(e).__name__, e)}
    # ---- BASELINE configs[3] on more ranks than samples: a TRIO on samples x chromosome groups (3 + 3 + 2 ranks at N = 8; sample_groups) — the multi-sample bin size and the bin
    # intersection span the world communicator, CanvasBin and PerSampleHMM run sharded inside each sample's sub-communicator (canvas_comm_split / canvas_comm_restore).  Rank 0
    # repeats the trio on its own GPU and compares its sample's result; the ranks of one sample must agree among themselves.  Untimed for `value`.
    grid = {}
    try:
        if world >= 3 and "error" not in part and "error" not in ped:
            from .lib import synth_generate_sample_device
            layout = sample_groups(world, 3)
            gs_, gg_, gn_ = layout[rank]
            gowner = owner_table(lengths, gn_)
            thr_g = torch.from_numpy(synth.poisson_thresholds(args.rate).view(np.int32)).to(device)
            gb, gm, gh = [None] * nchr, [None] * nchr, [None] * nchr
            for c in range(nchr):
                if gowner[c] == gg_:
                    if rank == 0: gb[c], gm[c] = cb[c], cm[c]                       # (rank 0's cohort sample was generated from `seed`: its bases / masks ARE the reference)
                    else: gb[c], _, gm[c], thr = synth_generate_device(seed, c, lengths[c], args.rate, device, thr)
                    gh[c] = synth_generate_sample_device(seed, seed + 5000 + 31 * gs_, c, int(lengths[c]), thr_g, device)[0]
            torch.cuda.synchronize()
            if _gloo():      # (the tests' one-GPU launch: host transport, one gloo group per sample)
                ggroups = [dist.new_group(ranks=[r_ for r_ in range(world) if layout[r_][0] == s_], backend="gloo") for s_ in range(3)]
                enter_world = lambda: init_host_comm(cv, rank, world)
                enter_group = lambda: init_host_comm(cv, gg_, gn_, group=ggroups[gs_])
            else:
                enter_world = lambda: restore_library_comm(cv)
                enter_group = lambda: split_library_comm(cv, gs_, gg_)
            flow = lambda: pedigree_grid_flow(cv, layout, rank, enter_world, enter_group, gb, gm, gh, lens, is_auto, flags)
            flow(); enter_world(); barrier()
            t0 = time.perf_counter(); gr = flow(); enter_world(); barrier()
            gsec = max_over_ranks(time.perf_counter() - t0, device)
            k = gr["n"]
            dig = torch.stack([gr["start"].to(torch.int64).sum(), gr["stop"].to(torch.int64).sum(), (gr["chr"].to(torch.int64) * (torch.arange(k, device=device) % 1009)).sum(),
                               torch.tensor(k, device=device), torch.tensor(int(gr["bin_size"]), device=device), gr["count"].view(torch.int32).to(torch.int64).sum(),
                               (gr["state"][:k].to(torch.int64) * (torch.arange(k, device=device) % 1013)).sum()])
            digs = all_gather_tensor(dig)
            # the bins (first five words) are the pedigree's: the same on every rank; counts and states are the sample's: the same inside a group
            same_bins = bool(all((d[:5] == digs[0][:5]).all() for d in digs))
            same_in_group = bool(all((digs[r_] == digs[rr_]).all() for r_ in range(world) for rr_ in range(world) if layout[r_][0] == layout[rr_][0]))
            eq = None; sec1 = None
            if rank == 0:
                t1 = time.perf_counter()
                allh = [[synth_generate_sample_device(seed, seed + 5000 + 31 * s_, c, int(lengths[c]), thr_g, device)[0] for c in range(nchr)] for s_ in range(3)]
                torch.cuda.synchronize()
                rates = []
                for s_ in range(3):
                    _, _, r_ = cv.bin_rates(allh[s_], cm, lens)
                    rates += [r_[c] for c in range(nchr) if is_auto[c]]
                bs1 = cv.bin_size_from_rates(rates, 100)
                outs1, tot1 = [], []
                for s_ in range(3):
                    o_ = dict(chr=mk(torch.int32), start=mk(torch.int32), stop=mk(torch.int32), gc=mk(torch.int32), count=mk(torch.float32))
                    _, _, t_ = cv.bin_genome(cb, cm, allh[s_], lens, bs1, 3, out=o_)
                    outs1.append(o_); tot1.append(int(t_))
                nout1, _, _ = cv.clean_batch(outs1, tot1, is_auto, flags)
                mc1, ms1, me1, mcnt1, k1 = cv.merge_cleaned(outs1, [int(x) for x in nout1])
                off1 = cv.chromosome_offsets(mc1, k1, nchr)
                st1 = cv.hmm_per_sample(cv.quantize_f2(mcnt1[gs_], k1), off1)
                sec1 = time.perf_counter() - t1
                eq = bool(bs1 == gr["bin_size"] and k1 == k and (ms1[:k1] == gr["start"]).all() and (me1[:k1] == gr["stop"]).all() and (mc1[:k1] == gr["chr"]).all()
                          and (mcnt1[gs_][:k1].view(torch.int32) == gr["count"].view(torch.int32)).all() and (st1[:k1] == gr["state"][:k1]).all())
                del allh, outs1
            grid = {"samples": 3, "ranks_per_sample": [int(sum(1 for l in layout if l[0] == s_)) for s_ in range(3)], "seconds": round(gsec, 4), "scaling": "strong (three samples, N ranks)",
                    "bin_size": int(gr["bin_size"]), "bins_common_to_all": int(k), "identical_on_all_ranks": bool(same_bins and same_in_group), "equals_single_gpu_flow_rank0": eq,
                    "single_gpu_seconds_incl_generating_the_samples_rank0": None if sec1 is None else round(sec1, 3),
                    "note": "a trio on samples x chromosome groups: rates all-gather (world) -> one bin size, canvas_bin_sample_sharded inside the sample's sub-communicator (canvas_comm_split), CanvasClean on "
                            "the group's ranks, canvas_merge_cleaned_sharded (world), canvas_hmm_per_sample_sharded (group)"}
            barrier()
    except Exception as e:                                        # noqa: BLE001
        grid = {"error": "%s: %s" % (type(e).__name__, e)}
        try: enter_world()
        except Exception: pass
    # ---- BASELINE configs[4], chromosomes sharded: tumour 80x (GCContentWeighted) + normal 40x of the owned chromosomes -> canvas_bin_sample_sharded x 2 -> ratio + CanvasClean on
    # every rank -> canvas_cbs_sharded.  Strong scaling (one pair, N ranks); rank 0 repeats the flow on its own GPU and compares.  Untimed for `value`.
    leg_done.set()
    som = {}
    som_done = threading.Event()

    def som_watchdog():            # this leg is reported, never fatal: if it hangs, the line goes out without it
        if not som_done.wait(float(os.environ.get("CANVAS_SHARDED_SOMATIC_TIMEOUT", "300"))):
            if rank == 0:
                result["pedigree_sharded"] = ped
                result["pedigree_grid"] = grid
                result["partition_sharded"] = part
                result["somatic_sharded"] = {"error": "did not finish within the watchdog's limit"}
                print(json.dumps(result), flush=True)
            os._exit(0)

    threading.Thread(target=som_watchdog, daemon=True).start()
    try:
        if "error" not in part and "error" not in ped and (world > 1 or os.environ.get("CANVAS_BENCH_FORCE_SHARDED")) and not os.environ.get("CANVAS_SHARDED_NO_SOMATIC"):
            from .lib import synth_generate_sample_device
            rt, rn = args.rate * 4.0 / 3.0, args.rate * 2.0 / 3.0
            thr_tt = torch.from_numpy(synth.poisson_thresholds(rt, purity=0.7).view(np.int32)).to(device)
            thr_nn = torch.from_numpy(synth.poisson_thresholds(rn, flat=True).view(np.int32)).to(device)
            def pair(which):
                ht, fl, hn = [None] * nchr, [None] * nchr, [None] * nchr
                for c in which:
                    ht[c], fl[c] = synth_generate_sample_device(seed, seed + 1000, c, int(lengths[c]), thr_tt, device, with_fraglen=True)
                    hn[c], _ = synth_generate_sample_device(seed, seed + 2000, c, int(lengths[c]), thr_nn, device)
                return ht, fl, hn
            mine = [c for c in range(nchr) if owner[c] == rank]
            ht, fl, hn = pair(mine)
            bb, mm = bases, masks                                  # the reference of the sharded sample: this rank's chromosomes (None elsewhere)
            torch.cuda.synchronize()
            call = lambda: cv.tumor_normal_flow(bb, mm, ht, fl, hn, lens, is_auto, flags, 0.01, 10000, owner=owner, keep=True)
            call(); barrier()
            cv.profile_get("allgather", reset=True)
            t0 = time.perf_counter(); sr = call(); barrier()
            ssec = max_over_ranks(time.perf_counter() - t0, device)
            ms_coll_s, k_coll_s = cv.profile_get("allgather")
            nc = int(sr["n_clean"])
            dig = torch.stack([sr["cov"][:nc].sum().to(torch.float64), sr["seg_len"].to(torch.float64).sum(), torch.tensor(float(nc), device=device, dtype=torch.float64),
                               torch.tensor(float(sr["bin_size"]), device=device, dtype=torch.float64), torch.tensor(float(int(sr["nseg"].sum())), device=device, dtype=torch.float64)])
            digs = all_gather_tensor(dig)
            same = bool(all((d == digs[0]).all() for d in digs))
            eq = None; sec1 = None
            if rank == 0:
                ht1, fl1, hn1 = pair(range(nchr))
                torch.cuda.synchronize()
                cv.tumor_normal_flow(cb, cm, ht1, fl1, hn1, lens, is_auto, flags, 0.01, 10000)      # (rank 0's cohort sample was generated from `seed`: its bases / masks ARE the reference)
                t1 = time.perf_counter(); one = cv.tumor_normal_flow(cb, cm, ht1, fl1, hn1, lens, is_auto, flags, 0.01, 10000, keep=True); sec1 = time.perf_counter() - t1
                eq = bool(one["bin_size"] == sr["bin_size"] and int(one["n_clean"]) == nc and (one["cov"][:nc] == sr["cov"][:nc]).all() and (np.asarray(one["nseg"]) == np.asarray(sr["nseg"])).all()
                          and (one["seg_len"] == sr["seg_len"]).all())
                del ht1, fl1, hn1
            som = {"seconds": round(ssec, 4), "single_gpu_seconds_rank0": None if sec1 is None else round(sec1, 4), "collectives": int(k_coll_s), "collective_ms": round(ms_coll_s, 4),
                   "redundant_clean_ms": (round(sr["stage_seconds"].get("clean", 0.0) * 1e3, 4) if isinstance(sr.get("stage_seconds"), dict) else None), "scaling": "strong", "bins": int(sr["n_bins"]), "bins_after_clean": nc,
                   "segments": int(sr["nseg"].sum()), "stage_seconds_rank0": sr["stage_seconds"] if rank == 0 else None, "identical_on_all_ranks": same, "equals_single_gpu_flow_rank0": eq,
                   "note": "one tumour / normal pair, chromosomes sharded: tumour bins -m GCContentWeighted (two small reductions + rate table + bin all-gather) and the normal's bins "
                           "through canvas_bin_sample_sharded, ratio + CanvasClean redundant, canvas_cbs_sharded"}
            barrier()
    except Exception as e:                                        # noqa: BLE001
        som = {"error": "%s: %s" % (type(e).__name__, e)}
    som_done.set()
    if rank == 0:
        result["pedigree_sharded"] = ped
        result["pedigree_grid"] = grid if grid else {"skipped": "needs at least three ranks (a trio)"}
        result["somatic_sharded"] = som
        part["note"] = "canvas_cbs_sharded / canvas_wavelets_sharded: every rank segments its own chromosomes (the reference's per-chromosome tasks), one list exchange; genome-wide inputs (seeds in file order, coverage variability) from the whole coverage on every rank"
        result["partition_sharded"] = part
        print(json.dumps(result), flush=True)
    dist.destroy_process_group()
    # a leg that raised, or whose result differs between the ranks / from the single-GPU result, fails the launch (the line above still says what happened)
    bad = "error" in part or "error" in ped                     # (the somatic and the grid legs are reported only: neither has ever run on more than one GPU)
    for leg in list(part.values()) + [ped]:
        if isinstance(leg, dict) and (leg.get("identical_on_all_ranks") is False or leg.get("equals_single_gpu_result") is False or leg.get("equals_single_gpu_flow_rank0") is False):
            bad = True
    if bad:
        sys.exit(6)
