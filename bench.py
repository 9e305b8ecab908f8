#!/usr/bin/env python3
"""bench.py — genome-bins/sec of the MI355X read-depth hot path (CanvasBin -> CanvasClean -> CanvasPartition).

One "step" = one pass of the hot path over one synthetic 60x whole-genome sample whose per-base arrays are already resident
in HBM (BASELINE.json configs[2]: GRCh38 chromosome lengths, 3.09e9 positions, hit rate ~0.21 => ~5.4 M bins):
    bin_rates -> bin size -> bin_genome -> clean (-g -s -r --local-sd-metric-file) -> F2 hand-off -> PerSampleHMM Viterbi
    -> segment ids [-> one RCCL all-gather of the per-rank boundary summary when N > 1].
N > 1: one process per GPU, each rank owns a different sample of the cohort (independent units, no data-path collective;
"scaling": "weak"); value = bins emitted by all ranks / max-over-ranks time.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant HBM kernel, hipEvent-timed inside the library on its own
stream) and "cpu_baseline" (the CPU oracle, timed on this box's host cores on a bounded sample; rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured float4 copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scale", type=float, default=1.0, help="fraction of each GRCh38 chromosome length (1.0 = BASELINE config)")
    ap.add_argument("--rate", type=float, default=0.21, help="hits per possible position (0.21 = 60x, 0.105 = 30x)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stage-times", action="store_true", help="print per-stage host wall times (ms) of the last step to stderr")
    ap.add_argument("--staged", action="store_true", help="time the six per-stage library calls from Python instead of the one-call canvas_sample_pipeline")
    ap.add_argument("--no-wavelets", action="store_true", help="skip the (untimed) Wavelets run on the cleaned coverage that is reported as wavelets_path")
    ap.add_argument("--no-cbs", action="store_true", help="skip the (untimed) CBS run on the cleaned coverage that is reported as cbs_path")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from canvas_amd import Canvas, synth, CLEAN_GCNORM, CLEAN_FILTSIZE, CLEAN_OUTLIERS, CLEAN_LOCALSD
    from canvas_amd.lib import synth_generate_device

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    cv = Canvas(local_rank)
    cv.profile_enable(True)
    if world > 1:
        import ctypes as C
        ident = [None]
        if rank == 0:
            buf = (C.c_ubyte * 128)()
            assert cv.lib.canvas_comm_unique_id(buf) == 0
            ident[0] = bytes(buf)
        dist.broadcast_object_list(ident, src=0)
        idbuf = (C.c_ubyte * 128).from_buffer_copy(ident[0])
        cv._check(cv.lib.canvas_comm_init(cv.ctx, rank, world, idbuf))

    # ---- synthetic sample resident in HBM
    from canvas_amd.parallel import sample_seed
    seed = sample_seed(20260927 + 3, rank)
    lengths = [max(200_000, int(L * args.scale)) for L in synth.GRCH38]
    nchr = len(lengths)
    lens = np.array(lengths, np.int64)
    thr = None
    bases, hits, masks = [], [], []
    for c, L in enumerate(lengths):
        b, h, m, thr = synth_generate_device(seed, c, L, args.rate, device, thr)
        bases.append(b); hits.append(h); masks.append(m)
    torch.cuda.synchronize()
    total_bases = int(lens.sum())
    is_auto = synth.IS_AUTOSOME
    flags = CLEAN_GCNORM | CLEAN_FILTSIZE | CLEAN_OUTLIERS | CLEAN_LOCALSD
    cap = int(total_bases // 100) + 16
    out = dict(chr=torch.empty(cap, dtype=torch.int32, device=device), start=torch.empty(cap, dtype=torch.int32, device=device),
               stop=torch.empty(cap, dtype=torch.int32, device=device), gc=torch.empty(cap, dtype=torch.int32, device=device),
               count=torch.empty(cap, dtype=torch.float32, device=device))
    cov_buf = torch.empty(cap, dtype=torch.float64, device=device)
    state_buf = torch.empty(cap, dtype=torch.int32, device=device)
    seg_buf = torch.empty(cap, dtype=torch.int32, device=device)
    gather_send = torch.zeros(4, dtype=torch.int32, device=device)
    gather_recv = torch.zeros(5 * world, dtype=torch.int32, device=device)
    keep = {}

    stage = {}

    def tick(name, t_prev):
        if args.stage_times:
            cv.synchronize(); torch.cuda.synchronize()
            now = time.perf_counter(); stage[name] = round((now - t_prev) * 1e3, 3); return now
        return t_prev

    def step(record=False):
        import ctypes as C
        if not record and not args.staged and not args.stage_times:
            # the whole path in ONE library call (canvas_sample_pipeline): same stages, no host-language overhead between them
            r = cv.sample_pipeline(bases, masks, hits, lens, is_auto, out, cov_buf, state_buf, seg_buf, counts_per_bin=100, bin_size=-1, mode=3, flags=flags,
                                   prepared=keep.get("prepared"))
            keep["prepared"] = r["prepared"]          # the marshalled pointer tables of the (unchanged) input arrays
            if world > 1:
                gather_send[0] = int(r["nseg"]); gather_send[1] = int(r["n_out"]); gather_send[2] = int(r["total"]); gather_send[3] = rank
                cnt = np.zeros(world, np.int32)
                cv._check(cv.lib.canvas_allgather_boundaries(cv.ctx, C.c_void_p(gather_send.data_ptr()), 4, 4, C.c_void_p(gather_recv.data_ptr()),
                                                             cnt.ctypes.data_as(C.c_void_p)))
            cv.synchronize()
            return r["total"]
        tp = time.perf_counter()
        o, per, total, bs = cv.bin_sample(bases, masks, hits, lens, is_auto, 100, -1, 3, out=out)    # CanvasBin -d 100 -m TruncatedDynamicRange
        tp = tick("bin_sample(rates+binning)", tp)
        if record:
            keep["binned"] = {k: v[:total].clone() for k, v in out.items()}
            tp = time.perf_counter()
        n_out, lsd, info = cv.clean(out, total, is_auto, flags)
        tp = tick("clean", tp)
        cov = cv.quantize_f2(out["count"], n_out, out=cov_buf)
        off_h = cv.chromosome_offsets(out["chr"], n_out, nchr)
        tp = tick("f2+offsets", tp)
        state = cv.hmm_per_sample(cov, off_h, out=state_buf)
        tp = tick("hmm", tp)
        seg, nseg = cv.segment_ids(off_h, state, out["start"], out["stop"], out=seg_buf)
        tp = tick("segment_ids", tp)
        if world > 1:
            gather_send[0] = int(nseg); gather_send[1] = int(n_out); gather_send[2] = int(total); gather_send[3] = rank
            cnt = np.zeros(world, np.int32)
            cv._check(cv.lib.canvas_allgather_boundaries(cv.ctx, C.c_void_p(gather_send.data_ptr()), 4, 4, C.c_void_p(gather_recv.data_ptr()),
                                                         cnt.ctypes.data_as(C.c_void_p)))
        cv.synchronize()
        if record:
            keep.update(bin_size=bs, total=total, n_out=n_out, lsd=lsd, info=info, cov=cov.clone(), off=off_h, state=state.clone(), seg=seg.clone(), nseg=nseg,
                        cleaned={k: v[:n_out].clone() for k, v in out.items()})
        return total

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    for name in ("bin_pass", "bin_tile_stats", "bin_summary", "bin_close", "viterbi", "viterbi_sequential", "viterbi_retry", "clean_total"):
        cv.profile_get(name, reset=True)
    barrier()
    t0 = time.perf_counter()
    bins_step = 0
    for i in range(args.steps):
        bins_step = step()
    barrier()
    dt = time.perf_counter() - t0
    step(record=True)        # untimed extra pass that keeps the intermediate arrays for the parity check / config fields
    from canvas_amd import parallel
    dt, total_bins_all, _ = parallel.aggregate_throughput(dt, float(bins_step), device=device)   # MAX over ranks, SUM of bins
    ms_per_step = dt / args.steps * 1e3
    value = total_bins_all / (dt / args.steps)

    # ---- roofline of the dominant HBM-bound kernel.  One-call binning reads the per-base arrays ONCE (k_tile_summary: 2.125 B/base read,
    # one 4-byte summary per 64 positions + 16 B per 4096-position tile written); the bins are then closed from the summaries (k_bin_close).
    # With CANVAS_BIN_TWO_PASS=1 the dominant kernel is k_bin_pass (2.125 B/base read + 16 B/bin written) after k_tile_stats (1.125 B/base).
    ms_bin, k_bin = cv.profile_get("bin_pass")
    ms_stats, k_stats = cv.profile_get("bin_tile_stats")
    ms_sum, k_sum = cv.profile_get("bin_summary")
    ms_close, k_close = cv.profile_get("bin_close")
    ms_vit, k_vit = cv.profile_get("viterbi")
    _, k_seq = cv.profile_get("viterbi_sequential")
    _, k_retry = cv.profile_get("viterbi_retry")
    ms_clean, k_clean = cv.profile_get("clean_total")
    clean_ms = ms_clean / max(1, k_clean)
    single_read = k_sum > 0
    if single_read:
        dom_kernel, pmc_name = "k_tile_summary", "pmc_tile_summary.json"
        ntiles = sum((int(L) + 4095) // 4096 for L in lens)
        alg_bytes = 2.125 * total_bases + 4.0 * ntiles * 64 + 16.0 * ntiles
        avg_ms, k_dom = ms_sum / max(1, k_sum), k_sum
        second = {"k_bin_close": {"avg_ms": round(ms_close / max(1, k_close), 4), "note": "reads the 64-position summaries (0.0625 B/base) and the 64 bases/hits under each bin boundary"}}
    else:
        dom_kernel, pmc_name = "k_bin_pass", "pmc_bin_pass.json"
        alg_bytes = 2.125 * total_bases + 16.0 * keep["total"]
        avg_ms, k_dom = ms_bin / max(1, k_bin), k_bin
        second = {"k_tile_stats": {"avg_ms": round(ms_stats / max(1, k_stats), 4),
                                   "achieved_GBs": round(1.125 * total_bases / max(1e-9, ms_stats / max(1, k_stats) * 1e-3) / 1e9, 1)}}
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic, traffic_src = None, None
    pmc = os.path.join(ROOT, "profiles", pmc_name)
    if os.path.exists(pmc):
        # PMC counters cannot be read from inside the timed run: FETCH_SIZE/WRITE_SIZE of this same deterministic workload were
        # collected with rocprofv3 --pmc in separate passes (tools/pmc_summary.py) and are quoted here per launch
        pj = json.load(open(pmc))
        if abs(pj["workload"]["scale"] - args.scale) < 1e-9 and abs(pj["workload"]["rate"] - args.rate) < 1e-9:
            traffic, traffic_src = pj["hbm_bytes_per_launch"], "profiles/" + pmc_name + " (rocprofv3 --pmc, FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)"
    roofline = {"kernel": dom_kernel, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src, "avg_ms": round(avg_ms, 4), "launches": k_dom,
                "algorithmic_bytes": alg_bytes,
                "other_kernels": {**second,
                                  "viterbi(speculate+backbone+verify)": {"avg_ms": round(ms_vit / max(1, k_vit), 4), "second_attempts": k_retry, "sequential_fallbacks": k_seq,
                                                                         "note": "recurrence-bound (16 B/bin algorithmic), not HBM-bound"},
                                  # CanvasClean is reported against the stage-sum 232 B/bin and the fused lower bound 32 B/bin;
                                  # the whole 5.4 M-bin SoA (150 MB) sits in the 256 MiB Infinity Cache, so the stage is launch/latency bound
                                  "canvas_clean(all stages)": {"avg_ms": round(clean_ms, 4),
                                                               "achieved_GBs_at_232B_per_bin": round(232.0 * keep["total"] / max(1e-9, clean_ms * 1e-3) / 1e9, 1),
                                                               "frac_of_peak_at_232B_per_bin": round(232.0 * keep["total"] / max(1e-9, clean_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                               "achieved_GBs_at_32B_per_bin": round(32.0 * keep["total"] / max(1e-9, clean_ms * 1e-3) / 1e9, 1)}}}

    result = {"metric": "genome-bins/sec (bin+clean+partition)", "value": round(value, 1), "unit": "bins/s", "n_gpus": world, "steps": args.steps,
              "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
              "dtype": "u8/int32 (bin), f32/f64 (clean, viterbi)", "data": "synthetic",
              "config": {"workload": "BASELINE configs[2]: whole-genome GRCh38 60x single sample, full bin+clean+partition HIP path on 1 MI355X per sample",
                         "bases_per_sample": total_bases, "bins_per_sample": int(keep["total"]), "bins_after_clean": int(keep["n_out"]), "bin_size": int(keep["bin_size"]),
                         "partition": "PerSampleHMM", "clean_flags": "-g -s -r --local-sd-metric-file", "segments": int(keep["nseg"]),
                         "samples": world, "scale": args.scale, "rate": args.rate},
              "roofline": roofline}

    if rank == 0 and world == 1 and not args.no_cbs:
        # the other partition method of the path (-m CBS, BASELINE configs[4]) on the same cleaned coverage; reported, not part of `value`
        t_c = time.perf_counter()
        cv.cbs(keep["cov"], keep["off"], 0.01, 10000)            # first call: sequential-boundary table (GetBoundary.cs), buffers, thread pool
        cbs_first = time.perf_counter() - t_c
        t_c = time.perf_counter()
        seg_len, nseg_c, cstats = cv.cbs(keep["cov"], keep["off"], 0.01, 10000)
        cbs_s = time.perf_counter() - t_c
        dstat = cv.cbs_device_stats()
        result["cbs_path"] = {"seconds": round(cbs_s, 3), "first_call_seconds": round(cbs_first, 3), "bins_per_s": round(int(keep["n_out"]) / cbs_s, 1), "segments": int(sum(nseg_c)), "tmaxo_calls": int(cstats[0]),
                              "permutations": int(cstats[2]), "permuted_elements": int(cstats[3]), "device_permutations": int(dstat[0]), "host_permutations": int(dstat[1]),
                              "exact_reevaluations": int(dstat[2]), "note": "CBSRunner.Run (alpha 0.01, 10000 permutations): recursion and stopping rule on the host, "
                              "TMaxO arc search + XPerm/HTMaxP + MT19937 on the device"}
    if rank == 0 and world == 1 and not args.no_wavelets:
        # the reference's default partition method (-m Wavelets) on the same cleaned coverage; reported, not part of `value`
        cv.profile_get("wavelet_chain", reset=True)
        t_w = time.perf_counter()
        bps = cv.wavelets(keep["cov"], keep["off"])
        wv_s = time.perf_counter() - t_w
        wst = cv.wavelets_stats()
        ms_chain, k_chain = cv.profile_get("wavelet_chain")
        wv = {"seconds": round(wv_s, 3), "bins_per_s": round(int(keep["n_out"]) / wv_s, 1), "breakpoints": int(sum(len(b) for b in bps)), "tree_levels": int(wst[0]),
              "chain_kernel_seconds": round(ms_chain / 1e3, 3), "nodes_recomputed_exactly": int(wst[1]),
              "note": "WaveletsRunner.Run (somatic flavour, default parameters): unbalanced Haar decomposition on the device level by level, "
                      "thresholding / healing on the host"}
        if not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            cov_h = keep["cov"].cpu().numpy(); off_h = keep["off"]
            per = [np.ascontiguousarray(cov_h[off_h[c]:off_h[c + 1]]) for c in range(len(off_h) - 1)]
            t_o = time.perf_counter()
            exp = O.wavelets_genome(per)
            wv["oracle_seconds_1_core"] = round(time.perf_counter() - t_o, 3)
            wv["parity_vs_oracle"] = bool(all(a.tolist() == b.tolist() for a, b in zip(bps, exp)))
        result["wavelets_path"] = wv
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(keep, bases, hits, masks, lens, is_auto, flags, total_bases)
    if rank == 0:
        if args.stage_times:
            print("stage times (ms, host wall incl. sync): " + json.dumps(stage), file=sys.stderr)
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(keep, bases, hits, masks, lens, is_auto, flags, total_bases):
    """The CPU oracle (oracle/, a restatement of the reference's algorithm — the C# original cannot be built here) timed on this
    box's cores on a bounded sample, and used at the same time as a full-size parity check of the GPU result."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    cores = os.cpu_count() or 1
    # Bin: the four smallest autosomes + their share of the rate pass, one thread per chromosome (Parallel.ForEach, CanvasBin.cs:539)
    sample = [18, 19, 20, 21]
    L = [int(lens[c]) for c in sample]
    hb = [bases[c][:l].cpu().numpy() for c, l in zip(sample, L)]
    hh = [hits[c][:l].cpu().numpy() for c, l in zip(sample, L)]
    hm = [masks[c].cpu().numpy().view(np.uint8) for c in sample]
    thr = min(cores, len(sample))
    t0 = time.perf_counter()
    O.bin_rates_genome(hm, hh, threads=thr)
    res = O.bin_genome(hb, hm, hh, keep["bin_size"], 3, threads=thr)
    t_bin_sample = time.perf_counter() - t0
    # parity of the sampled chromosomes' bins
    binned = {k: v.cpu().numpy() for k, v in keep["binned"].items()}
    ok_bins = True
    for i, c in enumerate(sample):
        sel = binned["chr"] == c
        ok_bins &= bool((binned["stop"][sel] == res[1][i]).all() and (binned["count"][sel] == res[3][i].astype(np.float32)).all() and (binned["gc"][sel] == res[2][i]).all())
    sample_bases = sum(L)
    # the reference runs one task per chromosome on all cores: extrapolate the sample's per-thread rate to the genome
    par = min(cores, 24)
    t_bin = t_bin_sample * thr / sample_bases * total_bases / par
    # Clean (single-threaded in the reference) and PerSampleHMM (one thread per chromosome) on ALL bins
    t0 = time.perf_counter()
    is_y = np.zeros(len(is_auto), np.uint8); is_y[-1] = 1
    ex = O.clean(binned["chr"], binned["start"], binned["stop"], binned["count"], binned["gc"], is_auto, is_y, flags)
    t_clean = time.perf_counter() - t0
    cleaned = {k: v.cpu().numpy() for k, v in keep["cleaned"].items()}
    ok_clean = bool(len(ex["chr"]) == keep["n_out"] and (ex["count"].view(np.uint32) == cleaned["count"].view(np.uint32)).all() and (ex["start"] == cleaned["start"]).all()
                    and ex["local_sd"] == keep["lsd"])
    cov = keep["cov"].cpu().numpy()
    off = keep["off"]
    per = [np.ascontiguousarray(cov[off[c]:off[c + 1]]) for c in range(len(off) - 1)]
    t0 = time.perf_counter()
    paths, ran = O.hmm_genome_per_sample(per, threads=par)
    t_hmm = time.perf_counter() - t0
    st = keep["state"].cpu().numpy()
    ok_states = bool((st == np.concatenate(paths)).all())
    t_total = t_bin + t_clean + t_hmm
    return {"value": round(keep["total"] / t_total, 1), "unit": "bins/s", "cores": par, "kind": "port",
            "sample": f"Bin: chr19-22 ({sample_bases} of {total_bases} bases) on {thr} threads, per-thread rate extrapolated to {par} threads; "
                      f"Clean (1 thread) and PerSampleHMM ({par} threads) on all {keep['total']} bins",
            "seconds": {"bin_sample": round(t_bin_sample, 3), "bin_extrapolated": round(t_bin, 3), "clean": round(t_clean, 3), "hmm": round(t_hmm, 3)},
            "parity_vs_gpu": {"bins_sampled_chromosomes": ok_bins, "clean_bitexact": ok_clean, "viterbi_states": ok_states},
            "note": "C++ restatement of the C# reference (which cannot be built here), -O2; a reported baseline, not a target"}


if __name__ == "__main__":
    main()
