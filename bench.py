#!/usr/bin/env python3
"""bench.py — genome-bins/sec of the MI355X read-depth hot path (CanvasBin -> CanvasClean -> CanvasPartition).

One "step" = one pass of the hot path over one synthetic 60x whole-genome sample (BASELINE.json configs[2]: GRCh38 chromosome lengths, 3.09e9 positions,
hit rate ~0.21 => ~4.8 M bins):
    bin_rates -> bin size -> bin_genome -> clean (-g -s -r --local-sd-metric-file) -> F2 hand-off -> PerSampleHMM Viterbi -> segment ids.
`value` is measured with the per-base arrays already resident in HBM when the timed region starts (the contract of the round).  SURVEY 8(d) / BASELINE.md
define the metric on the device region INCLUDING H2D/D2H: that figure is measured too, in its own timed region, and printed as `value_incl_h2d` (pinned host
arrays -> per-chromosome uploads on a copy stream overlapped with the sweep -> results copied back); `h2d` holds its break-down.

N > 1 (launched by torch.distributed.run, one rank per GPU):
    --multi sharded (default)  ONE sample, chromosomes sharded over the ranks (north_star / SURVEY 8e): local sweep -> all-gather of the per-chromosome rate
                               pairs -> one bin size -> local bins -> all-gather of the bins -> redundant deterministic CanvasClean -> PerSampleHMM on the owned
                               chromosomes -> ONE RCCL all-gather of the segment boundaries -> global segment ids.  "scaling": "strong".
    --multi cohort             one sample per rank, no data-path collective.  "scaling": "weak".

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant HBM kernel, hipEvent-timed inside the library on its own stream), "cpu_baseline" (the CPU
oracle on this box's host cores, whole genome, no extrapolation; rank 0, N = 1 only), "h2d", "packed_path", "cbs_path", "wavelets_path", "somatic_flow" (BASELINE configs[4]), "pedigree_flow" (BASELINE configs[3]).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # read by the HIP runtime at its first call: CBS keeps more than 4 kernels in flight.  8, not 16 (round 6): every hardware queue the
                                                    # runtime creates costs ~5 ms at the first streams and again at exit — 16: first CBS call 0.27-0.32 s, 8: 0.17-0.21, 4: 0.12; the warm tumour / normal CBS 0.39-0.40 / 0.40-0.41 / 0.41 s

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured float4 copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scale", type=float, default=1.0, help="fraction of each GRCh38 chromosome length (1.0 = BASELINE config)")
    ap.add_argument("--rate", type=float, default=0.21, help="hits per possible position (0.21 = 60x, 0.105 = 30x)")
    ap.add_argument("--multi", choices=["sharded", "cohort"], default="sharded", help="N > 1: one sample sharded by chromosome (strong scaling) or one sample per rank (weak)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-h2d", action="store_true", help="skip the H2D/D2H-inclusive timed region (value_incl_h2d)")
    ap.add_argument("--no-packed", action="store_true", help="skip the packed-planes leg (packed_path: 0.75 B/base inputs, HBM-resident and H2D-inclusive)")
    ap.add_argument("--stage-times", action="store_true", help="print per-stage host wall times (ms) of the last step to stderr")
    ap.add_argument("--staged", action="store_true", help="time the six per-stage library calls from Python instead of the one-call canvas_sample_pipeline")
    ap.add_argument("--no-wavelets", action="store_true", help="skip the (untimed) Wavelets run on the cleaned coverage that is reported as wavelets_path")
    ap.add_argument("--no-gc-only", action="store_true", help="skip the BASELINE configs[1] leg (30x sample, CanvasClean -g only, 20 B/bin accounting)")
    ap.add_argument("--no-executables", action="store_true", help="skip the (untimed for `value`) run of the three drop-in executables on files: the file-I/O-inclusive figure of SURVEY 8(d)")
    ap.add_argument("--no-cbs", action="store_true", help="skip the (untimed) CBS run on the cleaned coverage that is reported as cbs_path")
    ap.add_argument("--no-pedigree", action="store_true", help="skip the trio flow of BASELINE configs[3] that is reported as pedigree_flow")
    ap.add_argument("--no-somatic", action="store_true", help="skip the tumour / normal flow of BASELINE configs[4] that is reported as somatic_flow")
    args = ap.parse_args()
    launch_ranks_if_asked(args)

    import torch
    import torch.distributed as dist
    from canvas_amd import Canvas, synth, CLEAN_GCNORM, CLEAN_FILTSIZE, CLEAN_OUTLIERS, CLEAN_LOCALSD
    from canvas_amd.lib import synth_generate_device

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if world != max(1, args.gpus):
        # the launcher's world and --gpus must say the same thing: a line that claims n_gpus = N must have been measured on N ranks
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d, or without a launcher: bench.py starts its own ranks)" % (args.gpus, world, args.gpus))
    one_gpu = ONE_GPU_HOOK()      # test hook: every rank on GPU 0, exchanges through the library's host transport (gloo) — RCCL refuses two ranks on one device
    if world > 1 and not one_gpu and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: --gpus %d needs %d visible GPUs, this box has %d" % (world, world, torch.cuda.device_count()))
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    force_sharded = bool(os.environ.get("CANVAS_BENCH_FORCE_SHARDED")) and "RANK" in os.environ      # test hook: the N > 1 code path on a one-rank communicator
    if world > 1 or force_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    cv = Canvas(local_rank)
    cv.profile_enable(True)
    if world == 1 and not (args.no_cbs and args.no_somatic):
        # what INTEGRATION.md tells a host to do right behind canvas_create (and CanvasPartition -m CBS does during its file read): the draw streams of CBS are constants of
        # the method — the library starts generating them, and creates its launchers' streams, on a thread of its own while the host is busy with something else
        cv.cbs_prefetch(len(synth.GRCH38), 16 << 20)
    if world > 1 or force_sharded:
        from canvas_amd import parallel
        if one_gpu:
            parallel.init_host_comm(cv, rank, world)
        else:
            parallel.init_library_comm(cv, rank, world)
        if args.multi == "sharded":
            return sharded_main(args, cv, rank, world, device)

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
    keep = {}
    stage = {}

    def tick(name, t_prev):
        if args.stage_times:
            cv.synchronize(); torch.cuda.synchronize()
            now = time.perf_counter(); stage[name] = round((now - t_prev) * 1e3, 3); return now
        return t_prev

    def step(record=False):
        if not record and not args.staged and not args.stage_times:
            # the whole path in ONE library call (canvas_sample_pipeline): same stages, no host-language overhead between them
            r = cv.sample_pipeline(bases, masks, hits, lens, is_auto, out, cov_buf, state_buf, seg_buf, counts_per_bin=100, bin_size=-1, mode=3, flags=flags,
                                   prepared=keep.get("prepared"))
            keep["prepared"] = r["prepared"]          # the marshalled pointer tables of the (unchanged) input arrays
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

    # the timed passes carry ONE hipEvent pair (around the dominant kernel, for `roofline`): every further scope costs two barrier packets on the stream (~10 us of a pass);
    # the other stages' device times come from untimed passes with every scope on, right after the timed region
    ALL_SCOPES = ("bin_pass", "bin_tile_stats", "bin_summary", "bin_close", "bin_resolve", "viterbi", "viterbi_sequential", "viterbi_retry", "clean_total", "pass_span")
    cv.profile_enable(2)
    for _ in range(args.warmup):
        step()
    for name in ALL_SCOPES:
        cv.profile_get(name, reset=True)
    barrier()
    t0 = time.perf_counter()
    bins_step = 0
    for i in range(args.steps):
        bins_step = step()
    barrier()
    dt = time.perf_counter() - t0
    ms_bin, k_bin = cv.profile_get("bin_pass")
    ms_stats, k_stats = cv.profile_get("bin_tile_stats")
    ms_sum, k_sum = cv.profile_get("bin_summary")
    ms_span, k_span = cv.profile_get("pass_span")      # the timed passes' own device spans (one event pair per pass, on the library's stream)
    cv.profile_enable(1)
    for name in ALL_SCOPES:
        cv.profile_get(name, reset=True)
    PROF_PASSES = 5
    for _ in range(PROF_PASSES):
        step()
    barrier()
    step(record=True)        # untimed extra pass that keeps the intermediate arrays for the parity check / config fields
    from canvas_amd import parallel
    dt, total_bins_all, _ = parallel.aggregate_throughput(dt, float(bins_step), device=device)   # MAX over ranks, SUM of bins
    ms_per_step = dt / args.steps * 1e3
    value = total_bins_all / (dt / args.steps)

    # ---- roofline of the dominant HBM-bound kernel.  One-call binning reads the per-base arrays ONCE (k_tile_summary: 2.125 B/base read,
    # one 4-byte summary per 64 positions + 16 B per 4096-position tile written); the bins are then closed from the summaries (k_bin_close2 / k_bin_resolve_fin).
    # With CANVAS_BIN_TWO_PASS=1 the dominant kernel is k_bin_pass (2.125 B/base read + 16 B/bin written) after k_tile_stats (1.125 B/base).
    ms_close, k_close = cv.profile_get("bin_close")
    ms_res, k_res = cv.profile_get("bin_resolve")
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
        second = {"k_bin_close2+k_bin_resolve_fin": {"avg_ms": round(ms_close / max(1, k_close) + ms_res / max(1, k_res), 4), "close_ms": round(ms_close / max(1, k_close), 4), "resolve_ms": round(ms_res / max(1, k_res), 4),
                                                     "note": "reads the 64-position summaries (0.0625 B/base), then three 128-byte lines per bin (mask word, 64 bases, 64 hits: HBM-bound on line fills, tools/line_probe.hip)"}}
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
            traffic, traffic_src = pj["hbm_bytes_per_launch"], "profiles/" + pmc_name + " (rocprofv3 --pmc, FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; git head %s)" % pj.get("git_head", "round 2")
    clean_obj = {"avg_ms": round(clean_ms, 4),
                 "achieved_GBs_at_232B_per_bin": round(232.0 * keep["total"] / max(1e-9, clean_ms * 1e-3) / 1e9, 1),
                 "frac_of_peak_at_232B_per_bin": round(232.0 * keep["total"] / max(1e-9, clean_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                 "achieved_GBs_at_32B_per_bin": round(32.0 * keep["total"] / max(1e-9, clean_ms * 1e-3) / 1e9, 1)}
    roofline = {"kernel": dom_kernel, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src, "avg_ms": round(avg_ms, 4), "launches": k_dom,
                "algorithmic_bytes": alg_bytes,
                "other_kernels": {**second,
                                  "viterbi(speculate+backbone+verify)": {"avg_ms": round(ms_vit / max(1, k_vit), 4), "second_attempts": k_retry, "sequential_fallbacks": k_seq,
                                                                         "note": "recurrence-bound (16 B/bin algorithmic), not HBM-bound"},
                                  # CanvasClean is reported against the stage-sum 232 B/bin and the fused lower bound 32 B/bin
                                  "canvas_clean(all stages)": clean_obj}}

    # scalar copies of the figures the review asks about, inside `roofline` (the driver's record keeps this object whole)
    # what the stage really moves per bin (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/pmc_round.sh): read from the counter file of this round's code
    CLEAN_COUNTER_BYTES_PER_BIN, clean_counter_src = 86.8, "profiles/r03_pmc_clean_batch.txt (40.5 B fetched + 46.3 B written per bin)"
    pcb = os.path.join(ROOT, "profiles", "pmc_clean_batch.json")
    if os.path.exists(pcb):
        pj2 = json.load(open(pcb)); CLEAN_COUNTER_BYTES_PER_BIN = float(pj2["bytes_per_bin"])
        clean_counter_src = "profiles/pmc_clean_batch.json (%.1f B fetched + %.1f B written per bin; git head %s)" % (pj2["fetched_bytes_per_bin"], pj2["written_bytes_per_bin"], pj2.get("git_head", "?"))
    roofline["clean_ms"] = round(clean_ms, 4)
    roofline["clean_frac_232"] = clean_obj["frac_of_peak_at_232B_per_bin"]
    roofline["clean_frac_counter_bytes"] = round(CLEAN_COUNTER_BYTES_PER_BIN * keep["total"] / max(1e-9, clean_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    roofline["clean_counter_bytes_source"] = clean_counter_src
    roofline["bin_tail_ms"] = round(ms_close / max(1, k_close) + ms_res / max(1, k_res), 4) if single_read else None
    roofline["viterbi_ms"] = round(ms_vit / max(1, k_vit), 4)
    roofline["stage_scopes_from"] = "%d untimed passes with every scope on, after the timed region (the timed passes carry only the dominant kernel's event pair)" % PROF_PASSES
    if k_span:
        # measured IN the timed run: the device span of a pass (event pair around canvas_sample_pipeline on its own stream) against the wall time per pass
        roofline["pass_span_ms"] = round(ms_span / k_span, 4)
        roofline["hand_over_us_per_pass"] = round(max(0.0, ms_per_step - ms_span / k_span) * 1e3, 1)
        roofline["hand_over_source"] = "ms_per_step - pass_span_ms of the %d timed passes (hipEvent pair around the whole call, library stream): the device time per pass that belongs to no pass" % k_span
    tl = os.path.join(ROOT, "profiles", "pass_timeline.json")
    if os.path.exists(tl):
        tj = json.load(open(tl))
        if abs(tj.get("scale", -1) - args.scale) < 1e-9 and abs(tj.get("rate", -1) - args.rate) < 1e-9:
            roofline["idle_us_per_pass"] = tj["idle_us"]; roofline["idle_source"] = "profiles/pass_timeline.json (rocprofv3 --kernel-trace of this command, git head %s: span %.0f us, busy %.0f us; the run's own figure is hand_over_us_per_pass)" % (tj.get("git_head", "?"), tj["span_us"], tj["busy_us"])

    result = {"metric": "genome-bins/sec (bin+clean+partition)", "value": round(value, 1), "unit": "bins/s", "n_gpus": world, "steps": args.steps,
              "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
              "dtype": "u8/int32 (bin), f32/f64 (clean, viterbi)", "data": "synthetic",
              "config": {"workload": "BASELINE configs[2]: whole-genome GRCh38 60x single sample, full bin+clean+partition HIP path on 1 MI355X per sample",
                         "bases_per_sample": total_bases, "bins_per_sample": int(keep["total"]), "bins_after_clean": int(keep["n_out"]), "bin_size": int(keep["bin_size"]),
                         "partition": "PerSampleHMM", "clean_flags": "-g -s -r --local-sd-metric-file", "segments": int(keep["nseg"]),
                         "samples": world, "scale": args.scale, "rate": args.rate, "multi": "cohort" if world > 1 else None},
              "roofline": roofline}

    if rank == 0 and world == 1:
        # CanvasClean on a cohort (canvas_clean_batch): B copies of this sample's bins, every copy on its own stream.  The single-sample stage is a chain of ~60
        # launches on 134 MB that sit in the Infinity Cache; with B chains in flight the launch latencies overlap and B x 134 MB stream from HBM
        B = 8
        nb = int(keep["total"])
        t_b = []
        for rep in range(3):
            copies = [{k: v.clone() for k, v in keep["binned"].items()} for _ in range(B)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nout_b, lsd_b, _ = cv.clean_batch(copies, [nb] * B, is_auto, flags)
            t_b.append(time.perf_counter() - t0)
        okb = bool(all(int(x) == int(keep["n_out"]) for x in nout_b) and all(float(x) == float(keep["lsd"]) for x in lsd_b)
                   and all((c["count"][:int(keep["n_out"])].view(torch.int32) == keep["cleaned"]["count"].view(torch.int32)).all() for c in copies))
        per = min(t_b[1:]) / B
        clean_obj["cohort_batch"] = {"samples_in_flight": B, "ms_per_sample": round(per * 1e3, 4), "achieved_GBs_at_232B_per_bin": round(232.0 * nb / per / 1e9, 1),
                                     "frac_of_peak_at_232B_per_bin": round(232.0 * nb / per / 1e9 / HBM_PEAK_GBS, 4), "identical_to_single_sample_result": okb,
                                     "note": "canvas_clean_batch: host wall of the whole call / B (the call returns after the last sample's results are back)"}
        copies = None
    host = None
    if rank == 0 and world == 1 and not (args.no_h2d and args.no_cpu_baseline):
        # the host's copy of the per-base arrays (what LoadIntermediateData leaves in memory, CanvasBin.cs:965-969), pinned: source of the H2D-inclusive
        # region and input of the CPU baseline
        t_pin = time.perf_counter()
        host = {k: [torch.empty(t.shape, dtype=t.dtype, pin_memory=True).copy_(t) for t in src] for k, src in (("bases", bases), ("masks", masks), ("hits", hits))}
        torch.cuda.synchronize()
        host["pin_seconds"] = time.perf_counter() - t_pin
    if rank == 0 and world == 1 and not args.no_h2d:
        result["h2d"] = h2d_region(args, cv, torch, host, bases, masks, hits, lens, is_auto, flags, out, cov_buf, state_buf, seg_buf, keep, total_bases)
        result["value_incl_h2d"] = result["h2d"]["value_incl_h2d"]
    if rank == 0 and world == 1 and not args.no_packed:
        result["packed_path"] = packed_region(args, cv, torch, host, bases, masks, hits, lens, is_auto, flags, out, cov_buf, state_buf, seg_buf, keep, total_bases)
        if "value_incl_h2d" in result["packed_path"]:
            result["value_incl_h2d_packed"] = result["packed_path"]["value_incl_h2d"]
        if "two_bit_wire_form" in result["packed_path"]:
            result["value_incl_h2d_packed_two_bit"] = result["packed_path"]["two_bit_wire_form"]["value_incl_h2d"]
    if rank == 0 and world == 1 and not args.no_cbs:
        # the other partition method of the path (-m CBS, BASELINE configs[4]) on the same cleaned coverage; reported, not part of `value`
        t_c = time.perf_counter()
        cv.cbs(keep["cov"], keep["off"], 0.01, 10000)            # first call: sequential-boundary table (GetBoundary.cs), buffers, thread pool
        cbs_first = time.perf_counter() - t_c
        cbs_runs = []
        for _ in range(3):                                       # host threads + launcher round trips: single calls scatter by +-30 %, so three are timed and the median is reported
            t_c = time.perf_counter()
            seg_len, nseg_c, cstats = cv.cbs(keep["cov"], keep["off"], 0.01, 10000)
            cbs_runs.append(time.perf_counter() - t_c)
        cbs_s = sorted(cbs_runs)[1]
        dstat = cv.cbs_device_stats(); tstat = cv.cbs_tailp_stats()
        cb = {"tailp_decided_on_device": int(tstat[0]), "tailp_recomputed_on_host": int(tstat[1]), "seconds": round(cbs_s, 3), "seconds_of_each_call": [round(x, 3) for x in cbs_runs], "first_call_seconds": round(cbs_first, 3), "bins_per_s": round(int(keep["n_out"]) / cbs_s, 1), "segments": int(sum(nseg_c)), "tmaxo_calls": int(cstats[0]),
              "permutations": int(cstats[2]), "permuted_elements": int(cstats[3]), "device_permutations": int(dstat[0]), "host_permutations": int(dstat[1]),
              "exact_reevaluations": int(dstat[2]), "prefetch_at_context_creation": True, "draws_read_out_of_the_cache_last_call": int(cv.cbs_cache_stats()[0]), "draws_generated_inside_batches_last_call": int(cv.cbs_cache_stats()[1]),
              "cache_GB": round(int(cv.cbs_cache_stats()[4]) / 1e9, 2), "note": "first_call_seconds: the first canvas_cbs of the process, in a context that called canvas_cbs_prefetch when it was created (INTEGRATION.md 4: the draw streams and the launchers' streams come up in the background); CBSRunner.Run (alpha 0.01, 10000 permutations): recursion, stopping rule and the edge tests (TPermP) on the host; TMaxO arc search, TailP series, MT19937 and every "
              "permutation of the reference distribution (XPerm + HTMaxP / TMaxP) on the device; seconds = median of three warm calls"}
        if not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            cores = min(os.cpu_count() or 1, 24)
            cov_h = keep["cov"].cpu().numpy(); off_h = keep["off"]
            per = [np.ascontiguousarray(cov_h[off_h[c]:off_h[c + 1]]) for c in range(len(off_h) - 1)]
            t_o = time.perf_counter()
            exp_seg, est = O.cbs_genome(per, 0.01, 10000, threads=cores)
            cb["oracle_seconds"] = round(time.perf_counter() - t_o, 3); cb["oracle_threads"] = cores
            got = seg_len.cpu().numpy()
            cb["parity_vs_oracle"] = bool(all(int(nseg_c[c]) == len(exp_seg[c]) and (got[off_h[c]:off_h[c] + nseg_c[c]] == exp_seg[c]).all() for c in range(len(per)))
                                          and int(cstats[0]) == int(est[0]) and int(cstats[2]) == int(est[2]) and int(cstats[4]) == int(est[4]))
        result["cbs_path"] = cb
    if rank == 0 and world == 1 and not args.no_wavelets:
        # the reference's default partition method (-m Wavelets) on the same cleaned coverage; reported, not part of `value`
        t_w = time.perf_counter()
        cv.wavelets(keep["cov"], keep["off"])                   # first call: pinned staging arena, workspace, side stream (the same convention as cbs_path)
        wv_first = time.perf_counter() - t_w
        wv_runs = []
        for _ in range(3):
            cv.profile_get("wavelet_chain", reset=True)
            t_w = time.perf_counter()
            bps = cv.wavelets(keep["cov"], keep["off"])
            wv_runs.append(time.perf_counter() - t_w)
        wv_s = sorted(wv_runs)[1]
        wst = cv.wavelets_stats(); wdec = cv.wavelets_decisions()
        ms_chain, k_chain = cv.profile_get("wavelet_chain")
        wv = {"seconds": round(wv_s, 3), "seconds_of_each_call": [round(x, 3) for x in wv_runs], "first_call_seconds": round(wv_first, 3), "bins_per_s": round(int(keep["n_out"]) / wv_s, 1), "breakpoints": int(sum(len(b) for b in bps)), "tree_levels": int(wst[0]),
              "long_nodes_decided_from_the_closed_form": wdec[0], "long_nodes_undecided_sent_to_the_exact_chain": wdec[1], "long_nodes_chained_for_their_coefficient": wdec[2],
              "chain_kernel_seconds": round(ms_chain / 1e3, 3), "nodes_recomputed_exactly": int(wst[1]),
              "note": "WaveletsRunner.Run (somatic flavour, default parameters): unbalanced Haar tree built on the device — arg-max of every long node decided from exact integer "
                      "prefix sums + a rounding-error bound of the reference's recurrences, exact chains only for undecided nodes and for coefficients that may survive HardThresh — "
                      "thresholding / healing on the host"}
        t_e = time.perf_counter()
        ev = cv.evenness_score(keep["cov"], keep["off"], 100000)
        wv["evenness_score"] = ev; wv["evenness_seconds"] = round(time.perf_counter() - t_e, 4)
        if not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            cov_h = keep["cov"].cpu().numpy(); off_h = keep["off"]
            per = [np.ascontiguousarray(cov_h[off_h[c]:off_h[c + 1]]) for c in range(len(off_h) - 1)]
            cores = min(os.cpu_count() or 1, 24)
            t_o = time.perf_counter()
            exp = O.wavelets_genome(per, threads=cores)          # one task per chromosome, as WaveletsRunner.Run (Parallel.ForEach, WaveletsRunner.cs:89-135): the convention of every oracle_seconds of this file
            wv["oracle_seconds"] = round(time.perf_counter() - t_o, 3); wv["oracle_threads"] = cores
            wv["parity_vs_oracle"] = bool(all(a.tolist() == b.tolist() for a, b in zip(bps, exp)))
            t_o = time.perf_counter()
            ev_o = O.evenness_score(per, 100000)
            wv["evenness_oracle_seconds"] = round(time.perf_counter() - t_o, 3)
            wv["evenness_parity"] = bool((ev is None and ev_o is None) or (ev is not None and ev_o is not None and np.float64(ev).tobytes() == np.float64(ev_o).tobytes()))
        result["wavelets_path"] = wv
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(keep, host, lens, is_auto, flags, total_bases)
        if "value_incl_h2d" in result:
            result["h2d"]["speedup_vs_cpu_baseline_incl_h2d"] = round(result["value_incl_h2d"] / result["cpu_baseline"]["value"], 2)
            result["h2d"]["speedup_vs_cpu_baseline_hbm_resident"] = round(result["value"] / result["cpu_baseline"]["value"], 2)
        if "packed_path" in result:
            for k in ("value", "value_incl_h2d", "value_incl_h2d_reference_resident", "value_incl_h2d_and_host_packing_of_the_hits"):
                if k in result["packed_path"]:
                    result["packed_path"]["speedup_vs_cpu_baseline_" + k] = round(result["packed_path"][k] / result["cpu_baseline"]["value"], 2)
            tb = result["packed_path"].get("two_bit_wire_form")
            if tb:
                for k in ("value_incl_h2d", "value_incl_h2d_reference_resident", "value_incl_h2d_and_host_packing_of_the_hits"):
                    tb["speedup_vs_cpu_baseline_" + k] = round(tb[k] / result["cpu_baseline"]["value"], 2)
    host = None
    if rank == 0 and world == 1 and not args.no_gc_only:
        result["clean_gc_only_30x"] = clean_gc_only_leg(args, cv, torch, seed, bases, masks, lens, is_auto, device)
    if rank == 0 and world == 1 and not args.no_executables:
        result["executables"] = executables_leg(args, bases, masks, hits, lens, keep, result.get("cpu_baseline"))
    if rank == 0 and world == 1 and not args.no_somatic:
        result["somatic_flow"] = somatic_flow(args, cv, torch, seed, bases, masks, lens, is_auto, flags, device)
    if rank == 0 and world == 1 and not args.no_pedigree:
        result["pedigree_flow"] = pedigree_flow(args, cv, torch, seed, bases, masks, lens, is_auto, flags, device)
    if rank == 0:
        if args.stage_times:
            print("stage times (ms, host wall incl. sync): " + json.dumps(stage), file=sys.stderr)
        # the figures of the other legs once more as scalars, LAST in the line (the driver keeps the tail of stdout)
        summ = {"ms_per_step": result["ms_per_step"], "value": result["value"], "roofline_frac": roofline["frac"], "clean_ms": roofline["clean_ms"], "clean_frac_232": roofline["clean_frac_232"]}
        for k in ("value_incl_h2d", "value_incl_h2d_packed", "value_incl_h2d_packed_two_bit"):
            if k in result: summ[k] = result[k]
        for leg, keys in (("cbs_path", ("seconds", "first_call_seconds", "oracle_seconds", "parity_vs_oracle", "host_permutations")), ("wavelets_path", ("seconds", "first_call_seconds", "oracle_seconds", "parity_vs_oracle")),
                          ("clean_gc_only_30x", ("avg_ms", "frac_of_peak_at_20B_per_bin", "parity_vs_oracle")), ("packed_path", ("value", "ms_per_step"))):
            if leg in result:
                for k in keys:
                    if k in result[leg]: summ[leg + "." + k] = result[leg][k]
        if "somatic_flow" in result:
            sf = result["somatic_flow"]
            for k in ("seconds", "first_call_seconds", "cbs_oracle_seconds", "cbs_parity_vs_oracle"):
                if k in sf: summ["somatic_flow." + k] = sf[k]
            if isinstance(sf.get("stage_seconds"), dict):
                for k, v in sf["stage_seconds"].items(): summ["somatic_flow.stage." + k] = v
        if "executables" in result:
            for k, v in result["executables"].items():
                if k.startswith("wall_seconds") or k == "bins_per_s_file_io_inclusive": summ["executables." + k] = v
        if "cpu_baseline" in result: summ["speedup_hbm_resident_vs_cpu_baseline"] = round(result["value"] / result["cpu_baseline"]["value"], 1)
        result["summary"] = summ
        print(json.dumps(result), flush=True)
    cv.close()              # streams, engines and workspaces go while the runtime is still up (not from __del__ at interpreter shutdown)
    if world > 1:
        dist.destroy_process_group()


def ONE_GPU_HOOK():
    return os.environ.get("CANVAS_BENCH_ONE_GPU") == "1"


def launch_ranks_if_asked(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks here (one process per GPU under torch.distributed.run, rendezvous on
    127.0.0.1) and leave with their exit status.  Under a launcher (WORLD_SIZE set) this is a no-op and main() checks that the two agree.  Refuses — non-zero exit,
    nothing measured — when the box has fewer than N GPUs, so that a line with n_gpus = N can only come from N devices (CANVAS_BENCH_ONE_GPU=1, the tests' hook, puts
    every rank on GPU 0 over the host transport instead)."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    import subprocess
    if not ONE_GPU_HOOK():
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            print("bench.py: --gpus %d needs %d visible GPUs, this box has %d; nothing was measured" % (args.gpus, args.gpus, have), file=sys.stderr)
            raise SystemExit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def h2d_region(args, cv, torch, host, bases, masks, hits, lens, is_auto, flags, out, cov_buf, state_buf, seg_buf, keep, total_bases):
    """The contract-defined region of SURVEY 8(d): device region INCLUDING H2D/D2H.  Per pass: the three per-base arrays leave pinned host memory chromosome by
    chromosome on the library's copy stream (canvas_upload_genome_begin), every chromosome is swept as soon as it has arrived, the rest of the path follows, and
    the result columns (cleaned bins, coverage, state, segment id) are copied back to pinned host arrays."""
    n_cap = int(keep["total"])
    res_host = {k: torch.empty(n_cap, dtype=v.dtype, pin_memory=True) for k, v in out.items()}
    res_host.update(cov=torch.empty(n_cap, dtype=torch.float64, pin_memory=True), state=torch.empty(n_cap, dtype=torch.int32, pin_memory=True), seg=torch.empty(n_cap, dtype=torch.int32, pin_memory=True))
    nbytes = sum(int(t.numel()) * t.element_size() for k in ("bases", "masks", "hits") for t in host[k])

    def one(hits_only):
        cv.upload_genome_begin(lens, None if hits_only else host["bases"], bases, None if hits_only else host["masks"], masks, host["hits"], hits)
        r = cv.sample_pipeline(bases, masks, hits, lens, is_auto, out, cov_buf, state_buf, seg_buf, counts_per_bin=100, bin_size=-1, mode=3, flags=flags, prepared=keep.get("prepared"))
        n = int(r["n_out"])
        for k in ("chr", "start", "stop", "gc", "count"):
            cv.memcpy_d2h(res_host[k], out[k], n * out[k].element_size())
        cv.memcpy_d2h(res_host["cov"], cov_buf, n * 8); cv.memcpy_d2h(res_host["state"], state_buf, n * 4); cv.memcpy_d2h(res_host["seg"], seg_buf, n * 4)
        return r

    def timed(hits_only, reps):
        one(hits_only)                                        # warm-up
        cv.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = one(hits_only)
        cv.synchronize(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps, r

    reps = max(1, min(args.steps, 3))
    # the bare upload, for the break-down: all chromosomes, nothing overlapped
    cv.upload_genome_begin(lens, host["bases"], bases, host["masks"], masks, host["hits"], hits); cv.upload_genome_wait()
    t0 = time.perf_counter()
    cv.upload_genome_begin(lens, host["bases"], bases, host["masks"], masks, host["hits"], hits); cv.upload_genome_wait()
    t_up = time.perf_counter() - t0
    t_all, r = timed(False, reps)
    t_hits, r2 = timed(True, reps)
    n = int(r["n_out"])
    ok = bool(int(r["total"]) == int(keep["total"]) and n == int(keep["n_out"]) and (res_host["seg"][:n] == keep["seg"][:n].cpu()).all() and (res_host["count"][:n] == keep["cleaned"]["count"][:n].cpu()).all())
    d2h_bytes = n * (5 * 4 + 8 + 4 + 4)
    return {"value_incl_h2d": round(int(r["total"]) / t_all, 1), "seconds_per_pass_incl_h2d": round(t_all, 5), "passes": reps,
            "h2d_bytes": nbytes, "h2d_seconds": round(t_up, 5), "h2d_GBs": round(nbytes / t_up / 1e9, 2), "d2h_bytes": d2h_bytes,
            "overlap": "per-chromosome uploads on a copy stream, each chromosome swept when it has arrived (pass ~ max(PCIe, compute) + tail)",
            "pass_minus_bare_upload_ms": round((t_all - t_up) * 1e3, 3),
            "value_incl_h2d_reference_resident": round(int(r2["total"]) / t_hits, 1), "seconds_per_pass_reference_resident": round(t_hits, 5),
            "reference_resident_note": "bases and possible-alignment mask are the same for every sample of a cohort (reference genome + kmer.fa): only the hit array (1 B/base) is uploaded",
            "results_identical_to_resident_path": ok, "pin_seconds": round(host["pin_seconds"], 2),
            "note": "SURVEY 8(d) / BASELINE.md define genome-bins/sec on the device region incl. H2D/D2H: value_incl_h2d is that figure (PCIe-bound: 2.125 B/base over the link); "
                    "`value` is the HBM-resident figure the round's contract asks for"}


def packed_region(args, cv, torch, host, bases, masks, hits, lens, is_auto, flags, out, cov_buf, state_buf, seg_buf, keep, total_bases):
    """The same pass over the packed planes (include/canvas_hip.h "packed per-base inputs": {possible, gc} bit pairs + bit-sliced 4-bit hit counters, 0.75 B/base
    instead of 2.125 B/base): HBM-resident rate, the sweep kernel against ITS bytes, and the H2D-inclusive region with the planes leaving pinned host memory.
    Reported beside `value` / `value_incl_h2d`, which stay on the reference's own in-memory arrays (byte per base)."""
    from canvas_amd.lib import pack_reference_host, pack_hits_host, packed_plane_words
    dref, dpl, pos0, sat = cv.pack_genome_device(bases, masks, hits, lens)
    kw = dict(counts_per_bin=100, bin_size=-1, mode=3, flags=flags, pos0=pos0)
    r = cv.sample_pipeline(dref, None, dpl, lens, is_auto, out, cov_buf, state_buf, seg_buf, **kw)
    prepared = r["prepared"]
    cv.synchronize()
    n = int(r["n_out"])
    same = bool(int(r["total"]) == int(keep["total"]) and n == int(keep["n_out"]) and int(r["nseg"]) == int(keep["nseg"]) and torch.equal(seg_buf[:n], keep["seg"][:n])
                and torch.equal(out["count"][:n], keep["cleaned"]["count"][:n]) and torch.equal(out["stop"][:n], keep["cleaned"]["stop"][:n]) and torch.equal(state_buf[:n], keep["state"][:n]))
    cv.profile_get("bin_summary_packed", reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        cv.sample_pipeline(None, None, None, None, None, None, None, None, None, prepared=prepared)
        cv.synchronize()
    torch.cuda.synchronize()
    t_res = (time.perf_counter() - t0) / args.steps
    ms_k, k_k = cv.profile_get("bin_summary_packed")
    ntiles = sum((int(L) + 4095) // 4096 for L in lens)
    alg = 48.0 * ntiles * 64 + 4.0 * ntiles * 64 + 16.0 * ntiles             # planes read, summaries + tile totals written
    avg_ms = ms_k / max(1, k_k)
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_tile_summary_packed.json")
    if os.path.exists(pmc):
        pj = json.load(open(pmc))
        if abs(pj["workload"]["scale"] - args.scale) < 1e-9 and abs(pj["workload"]["rate"] - args.rate) < 1e-9:
            traffic = pj["hbm_bytes_per_launch"]
    res = {"value": round(int(r["total"]) / t_res, 1), "ms_per_step": round(t_res * 1e3, 3), "steps": args.steps, "identical_to_byte_array_path": same,
           "input_bytes_per_base": 0.75, "saturated_hit_positions": int(sat),
           "roofline": {"kernel": "k_tile_summary_packed", "bound": "hbm", "achieved": round(alg / max(1e-9, avg_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(alg / max(1e-9, avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic, "avg_ms": round(avg_ms, 4), "launches": k_k, "algorithmic_bytes": alg,
                        "traffic_source": "profiles/pmc_tile_summary_packed.json (rocprofv3 --pmc, FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)" if traffic else None,
                        "note": "48 B read per 64 positions (0.75 B/base) + 4 B summary per 64 positions + 16 B per tile written; the byte-array sweep moves 6.77 GB for the same result"},
           "note": "same pass, same results; inputs = reference planes {possible, gc} (16 B / 64 positions, per reference genome) + hit planes (bit-sliced min(15, hits), 32 B / 64 positions)"}
    # several samples of a cohort in flight on ONE GPU (one context + host thread per sample over the same resident reference planes): the latency-bound stages of
    # one sample fill the gaps of another's
    import threading
    from canvas_amd import Canvas
    S = 4
    ctxs, preps, bufs = [], [], []
    for i in range(S):
        c2 = Canvas(cv.device.index)
        mk = lambda dt: torch.empty(out["chr"].numel(), dtype=dt, device=cv.device)
        o2 = dict(chr=mk(torch.int32), start=mk(torch.int32), stop=mk(torch.int32), gc=mk(torch.int32), count=mk(torch.float32))
        b2 = (o2, mk(torch.float64), mk(torch.int32), mk(torch.int32))
        r2 = c2.sample_pipeline(dref, None, dpl, lens, is_auto, b2[0], b2[1], b2[2], b2[3], **kw)
        c2.synchronize()
        ctxs.append(c2); preps.append(r2["prepared"]); bufs.append(b2)

    def run(i, k):
        for _ in range(k):
            ctxs[i].sample_pipeline(None, None, None, None, None, None, None, None, None, prepared=preps[i])
        ctxs[i].synchronize()

    k_each = max(3, args.steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(i, k_each)) for i in range(S)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    t_fl = (time.perf_counter() - t0) / (S * k_each)
    same_fl = bool(all(torch.equal(b[3][:n], keep["seg"][:n]) and torch.equal(b[0]["count"][:n], keep["cleaned"]["count"][:n]) for b in bufs))
    res["samples_in_flight"] = {"samples": S, "passes_each": k_each, "ms_per_sample": round(t_fl * 1e3, 3), "value": round(int(r["total"]) / t_fl, 1), "identical_results": same_fl,
                                "note": "cohort on one GPU: S contexts, one host thread each, same resident reference planes, each sample's own hit planes would differ in production"}
    for c2 in ctxs:
        c2.close()
    ctxs = preps = bufs = None
    if host is not None and not args.no_h2d:
        cores = min(os.cpu_count() or 1, 24)
        href = [torch.empty(2 * packed_plane_words(L), dtype=torch.int64, pin_memory=True) for L in lens]
        hpl = [torch.empty(4 * packed_plane_words(L), dtype=torch.int64, pin_memory=True) for L in lens]
        t0 = time.perf_counter()
        hp0 = [pack_reference_host(host["bases"][c], host["masks"][c], int(lens[c]), out=href[c], threads=cores)[1] for c in range(len(lens))]
        t_pref = time.perf_counter() - t0
        t0 = time.perf_counter()
        for c in range(len(lens)):
            pack_hits_host(host["hits"][c], int(lens[c]), out=hpl[c], threads=cores)
        t_phit = time.perf_counter() - t0
        packers_agree = bool(list(hp0) == list(pos0) and all(torch.equal(a, b.cpu()) for a, b in zip(href[:3], dref[:3])) and all(torch.equal(a, b.cpu()) for a, b in zip(hpl[:3], dpl[:3])))
        n_cap = int(keep["total"])
        res_host = {k: torch.empty(n_cap, dtype=v.dtype, pin_memory=True) for k, v in out.items()}
        res_host.update(cov=torch.empty(n_cap, dtype=torch.float64, pin_memory=True), state=torch.empty(n_cap, dtype=torch.int32, pin_memory=True), seg=torch.empty(n_cap, dtype=torch.int32, pin_memory=True))

        def one(hits_only):
            cv.upload_packed_begin(lens, None if hits_only else href, dref, hpl, dpl)
            rr = cv.sample_pipeline(None, None, None, None, None, None, None, None, None, prepared=prepared)
            m = int(rr["n_out"])
            for k in ("chr", "start", "stop", "gc", "count"):
                cv.memcpy_d2h(res_host[k], out[k], m * out[k].element_size())
            cv.memcpy_d2h(res_host["cov"], cov_buf, m * 8); cv.memcpy_d2h(res_host["state"], state_buf, m * 4); cv.memcpy_d2h(res_host["seg"], seg_buf, m * 4)
            return rr

        def timed(hits_only, reps):
            one(hits_only)
            cv.synchronize(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                rr = one(hits_only)
            cv.synchronize(); torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps, rr

        reps = max(1, min(args.steps, 3))
        for t in dpl:
            t.zero_()                                              # the planes on the device really come from the host copies below
        t_all, ra = timed(False, reps)
        t_hits, rb = timed(True, reps)
        m = int(ra["n_out"])
        ok = bool(int(ra["total"]) == int(keep["total"]) and m == int(keep["n_out"]) and (res_host["seg"][:m] == keep["seg"][:m].cpu()).all()
                  and (res_host["count"][:m] == keep["cleaned"]["count"][:m].cpu()).all())
        nbytes = sum(int(t.numel()) * 8 for t in href + hpl)
        # a cohort through the same region: two samples in flight (a context, a copy stream and a host thread each, the reference planes resident and shared), so one sample's
        # tail (the rest of the pass after the last chromosome has arrived, the results on their way back) runs under the other's upload: throughput -> hit planes / PCIe
        S2 = 2
        lanes = []
        for i in range(S2):
            c2 = Canvas(cv.device.index)
            mk = lambda dt: torch.empty(out["chr"].numel(), dtype=dt, device=cv.device)
            o2 = dict(chr=mk(torch.int32), start=mk(torch.int32), stop=mk(torch.int32), gc=mk(torch.int32), count=mk(torch.float32))
            cb2, sb2, gb2 = mk(torch.float64), mk(torch.int32), mk(torch.int32)
            dpl2 = [torch.zeros_like(t) for t in dpl]
            rh2 = {k: torch.empty(n_cap, dtype=v.dtype, pin_memory=True) for k, v in o2.items()}
            rh2.update(cov=torch.empty(n_cap, dtype=torch.float64, pin_memory=True), state=torch.empty(n_cap, dtype=torch.int32, pin_memory=True), seg=torch.empty(n_cap, dtype=torch.int32, pin_memory=True))
            c2.upload_packed_begin(lens, None, dref, hpl, dpl2)
            r2 = c2.sample_pipeline(dref, None, dpl2, lens, is_auto, o2, cb2, sb2, gb2, **kw)
            c2.synchronize()
            lanes.append(dict(cv=c2, prep=r2["prepared"], out=o2, cov=cb2, state=sb2, seg=gb2, dpl=dpl2, host=rh2, n=int(r2["n_out"])))

        def lane_run(L, k):
            c2 = L["cv"]
            for _ in range(k):
                c2.upload_packed_begin(lens, None, dref, hpl, L["dpl"])
                rr = c2.sample_pipeline(None, None, None, None, None, None, None, None, None, prepared=L["prep"])
                m2 = int(rr["n_out"])
                for kk in ("chr", "start", "stop", "gc", "count"):
                    c2.memcpy_d2h(L["host"][kk], L["out"][kk], m2 * L["out"][kk].element_size())
                c2.memcpy_d2h(L["host"]["cov"], L["cov"], m2 * 8); c2.memcpy_d2h(L["host"]["state"], L["state"], m2 * 4); c2.memcpy_d2h(L["host"]["seg"], L["seg"], m2 * 4)
                c2.synchronize()

        k_each = max(2, reps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=lane_run, args=(L, k_each)) for L in lanes]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        t_pipe = (time.perf_counter() - t0) / (S2 * k_each)
        ok_pipe = bool(all(L["n"] == int(keep["n_out"]) and (L["host"]["seg"][:L["n"]] == keep["seg"][:L["n"]].cpu()).all() for L in lanes))
        for L in lanes:
            L["cv"].close()
        lanes = None
        res["cohort_incl_h2d_two_samples_in_flight"] = {"value": round(int(rb["total"]) / t_pipe, 1), "seconds_per_sample": round(t_pipe, 5), "samples_in_flight": S2, "passes_each": k_each,
                                                         "results_identical": ok_pipe,
                                                         "note": "reference planes resident; per sample 1.54 GB of hit planes up, 168 MB of results back; the two samples' uploads share the link"}
        # the hit planes in their two-bit wire form (0.25 B/base + the few words with four hits and more), expanded on the device behind each chromosome's transfer
        from canvas_amd.lib import pack_hits2_host
        W = [packed_plane_words(L) for L in lens]
        h_lo = [torch.empty(2 * w, dtype=torch.int64, pin_memory=True) for w in W]; h_hdr = [torch.empty(2 * (w // 64), dtype=torch.int64, pin_memory=True) for w in W]
        h_ex = [torch.empty(2 * (w // 16 + 64), dtype=torch.int64, pin_memory=True) for w in W]
        t0 = time.perf_counter()
        nx = [pack_hits2_host(host["hits"][c], int(lens[c]), lo=h_lo[c], hdr=h_hdr[c], extras=h_ex[c], threads=cores)[3] for c in range(len(lens))]
        t_p2 = time.perf_counter() - t0

        def one2(hits_only):
            cv.upload_packed2_begin(lens, None if hits_only else href, dref, h_lo, h_hdr, h_ex, nx, dpl)
            rr = cv.sample_pipeline(None, None, None, None, None, None, None, None, None, prepared=prepared)
            m2 = int(rr["n_out"])
            for k in ("chr", "start", "stop", "gc", "count"):
                cv.memcpy_d2h(res_host[k], out[k], m2 * out[k].element_size())
            cv.memcpy_d2h(res_host["cov"], cov_buf, m2 * 8); cv.memcpy_d2h(res_host["state"], state_buf, m2 * 4); cv.memcpy_d2h(res_host["seg"], seg_buf, m2 * 4)
            return rr

        def timed2(hits_only):
            for t in dpl:
                t.zero_()
            one2(hits_only)
            cv.synchronize(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                rr = one2(hits_only)
            cv.synchronize(); torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps, rr

        t2_all, r2a = timed2(False)
        t2_hits, r2b = timed2(True)
        m2 = int(r2b["n_out"])
        ok2 = bool(int(r2b["total"]) == int(keep["total"]) and m2 == int(keep["n_out"]) and (res_host["seg"][:m2] == keep["seg"][:m2].cpu()).all()
                   and (res_host["count"][:m2] == keep["cleaned"]["count"][:m2].cpu()).all())
        b2 = sum(int(t.numel()) * 8 for t in h_lo + h_hdr) + 16 * int(sum(nx))
        res["two_bit_wire_form"] = {"value_incl_h2d": round(int(r2a["total"]) / t2_all, 1), "seconds_per_pass_incl_h2d": round(t2_all, 5),
                                    "value_incl_h2d_reference_resident": round(int(r2b["total"]) / t2_hits, 1), "seconds_per_pass_reference_resident": round(t2_hits, 5),
                                    "hit_plane_bytes_over_pcie": b2, "words_with_four_hits_and_more": int(sum(nx)), "results_identical": ok2,
                                    "host_pack_seconds_from_byte_array": round(t_p2, 4),
                                    "value_incl_h2d_and_host_packing_of_the_hits": round(int(r2b["total"]) / (t2_hits + t_p2), 1),
                                    "note": "hit planes as {b0, b1} per word + a per-tile header + the {b2, b3} of the words that have them (canvas_pack_hits2_host / canvas_upload_packed2_begin)"}
        res.update({"value_incl_h2d": round(int(ra["total"]) / t_all, 1), "seconds_per_pass_incl_h2d": round(t_all, 5), "h2d_bytes": nbytes,
                    "value_incl_h2d_reference_resident": round(int(rb["total"]) / t_hits, 1), "seconds_per_pass_reference_resident": round(t_hits, 5),
                    "h2d_results_identical": ok, "host_packers_agree_with_device_packer": packers_agree,
                    "host_pack_seconds": {"reference_planes": round(t_pref, 4), "hit_planes": round(t_phit, 4), "threads": cores,
                                          "note": "canvas_pack_reference_host (once per reference genome) / canvas_pack_hits_host (once per sample) from the byte arrays; "
                                                  "a host that fills the planes while it parses the BAM pays neither"},
                    "value_incl_h2d_and_host_packing_of_the_hits": round(int(rb["total"]) / (t_hits + t_phit), 1)})
    return res


def clean_gc_only_leg(args, cv, torch, seed, bases, masks, lens, is_auto, device):
    """BASELINE configs[1]: whole-genome 30x single sample, CanvasClean's GC normalisation alone (-g: RemoveBinsWithExtremeGC + NormalizeByGC, CanvasClean.cs:163-237) on one
    MI355X, against SURVEY 8(d)'s 20 B/bin (statistics pass 8 B, apply pass 8 + 4 B).  The bins come from CanvasBin on a 30x sample over the same reference (half the hit
    rate of the headline sample): the stage is timed by the library's own hipEvent scope, single sample and cohort of 8."""
    from canvas_amd import synth, CLEAN_GCNORM
    from canvas_amd.lib import synth_generate_sample_device
    thr = torch.from_numpy(synth.poisson_thresholds(args.rate / 2.0).view(np.int32)).to(device)
    hits30 = [synth_generate_sample_device(seed, seed + 3000, c, int(L), thr, device)[0] for c, L in enumerate(lens)]
    torch.cuda.synchronize()             # (the generator runs on torch's stream, the library on its own)
    cap = int(int(lens.sum()) // 100) + 16
    mk = lambda dt: torch.empty(cap, dtype=dt, device=device)
    out = dict(chr=mk(torch.int32), start=mk(torch.int32), stop=mk(torch.int32), gc=mk(torch.int32), count=mk(torch.float32))
    _, per, total, bs = cv.bin_sample(bases, masks, hits30, lens, is_auto, 100, -1, 3, out=out)
    del hits30
    binned = {k: v[:total].clone() for k, v in out.items()}
    work = {k: v.clone() for k, v in binned.items()}
    cv.profile_enable(True); cv.profile_get("clean_total", reset=True)
    reps = 5; n_out = 0
    for r in range(reps + 1):
        for k in work:
            work[k].copy_(binned[k])
        torch.cuda.synchronize()
        if r == 1:
            cv.profile_get("clean_total", reset=True)
        n_out, _, info = cv.clean(work, total, is_auto, CLEAN_GCNORM)
    ms, k = cv.profile_get("clean_total")
    ms = ms / max(1, k)
    o = {"workload": "BASELINE configs[1]: whole-genome 30x single sample (rate %.4f), CanvasClean -g only" % (args.rate / 2.0), "bins": int(total), "bin_size": int(bs), "bins_after": int(n_out),
         "avg_ms": round(ms, 4), "achieved_GBs_at_20B_per_bin": round(20.0 * total / (ms * 1e-3) / 1e9, 1), "frac_of_peak_at_20B_per_bin": round(20.0 * total / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
         "counting_selects": bool(info[5]),
         "three_launch_stage": bool(info[6]),
         "note": "clean_gc_only.hpp: three launches, in place (k_cg_count: per-workgroup (GC, count) counters in LDS written as slabs; k_cg_medians: every workgroup takes the "
                 "strip decision for itself, one workgroup per bucket reads its median off the summed rows; k_cg_apply: register-held apply + strip with a chunk hand-off). "
                 "It moves 12 B/bin (statistics incl. the chromosome index) + 40 B/bin (all five columns are read and rewritten: the strip shifts every bin behind the first "
                 "stripped one) + 5 B/bin of slabs against the contract's 20 B/bin, and each launch has a ~2.5 us floor"}
    B = 8
    best = None
    for r in range(3):
        copies = [{k: v.clone() for k, v in binned.items()} for _ in range(B)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nout_b, _, _ = cv.clean_batch(copies, [total] * B, is_auto, CLEAN_GCNORM)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    same = bool(all(int(x) == int(n_out) for x in nout_b) and all(torch.equal(c["count"][:int(n_out)].view(torch.int32), work["count"][:int(n_out)].view(torch.int32)) for c in copies))
    per_s = best / B
    o["cohort_batch"] = {"samples_in_flight": B, "ms_per_sample": round(per_s * 1e3, 4), "achieved_GBs_at_20B_per_bin": round(20.0 * total / per_s / 1e9, 1),
                         "frac_of_peak_at_20B_per_bin": round(20.0 * total / per_s / 1e9 / HBM_PEAK_GBS, 4), "identical_to_single_sample_result": same,
                         "note": "host wall of the whole canvas_clean_batch call / B"}
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        h = {k: v.cpu().numpy() for k, v in binned.items()}
        t0 = time.perf_counter()
        ex = O.clean(h["chr"], h["start"], h["stop"], h["count"], h["gc"], is_auto, np.zeros(len(is_auto), np.uint8), CLEAN_GCNORM)
        o["oracle_seconds"] = round(time.perf_counter() - t0, 3); o["oracle_threads"] = 1
        o["parity_vs_oracle"] = bool(len(ex["chr"]) == int(n_out) and (ex["count"].view(np.uint32) == work["count"][:int(n_out)].cpu().numpy().view(np.uint32)).all()
                                     and (ex["start"] == work["start"][:int(n_out)].cpu().numpy()).all())
    return o


def _varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80); v >>= 7
    out.append(v)
    return bytes(out)


def _ld_header(field, nbytes):
    return _varint(field << 3 | 2) + _varint(nbytes)


def _write_dat(path, name, mask_bytes, hits, L):
    """CanvasBin.IntermediateData of one chromosome as protobuf-net writes it (CanvasBin.cs:1037-1072; the layout tests/test_canvasbin_tool_gpu.py pins):
    member 1 possible-alignment bits (read back least significant bit first), member 2 observed alignments, member 3 bits in the last byte"""
    key = _ld_header(1, len(name)) + name.encode()
    with open(path, "wb") as f:
        for field, payload in ((1, mask_bytes), (2, hits)):
            val = _ld_header(2, len(payload))
            f.write(_ld_header(field, len(key) + len(val) + len(payload)) + key + val)
            payload.tofile(f)
        last = _varint(2 << 3 | 0) + _varint(int(L % 8))
        f.write(_ld_header(3, len(key) + len(last)) + key + last)


def executables_leg(args, bases, masks, hits, lens, keep, cpu):
    """What a user of the reference launches (CanvasRunner.cs:123-128): the three drop-in executables exchanging files.  The synthetic sample is written once as kmer.fa +
    per-chromosome .dat intermediates (CanvasBin -c's output), then CanvasBin -i -> CanvasClean -g -s -r --local-sd-metric-file -> CanvasPartition -m {PerSampleHMM, CBS,
    Wavelets} run as processes, wall-clock, file I/O (gzip text, protobuf, FASTA) and host-side packing included; every tool reports how its wall time splits into reading,
    device work (incl. H2D / D2H and context creation) and writing.  The CPU column is an ESTIMATE: the same tools with the device phase replaced by the oracle's seconds on
    24 threads (cpu_baseline) — the reference's C# cannot be built here, and its file I/O is the same work."""
    import shutil, subprocess, tempfile
    from canvas_amd import synth
    root = tempfile.mkdtemp(prefix="canvas_exe_", dir=os.environ.get("TMPDIR", "/tmp"))
    keep_dir = os.environ.get("CANVAS_EXE_KEEP")            # experiments on the tools: the sample's files stay, the CanvasBin command line is written next to them
    try:
        total_bytes = int(sum(int(L) for L in lens))
        free = shutil.disk_usage(root).free
        nchr = len(lens)
        if free < 2.6 * total_bytes + (2 << 30):
            return {"skipped": "not enough space under %s for the sample's files (%.1f GB free, %.1f GB needed)" % (root, free / 1e9, (2.6 * total_bytes + (2 << 30)) / 1e9)}
        names = synth.CHROM_NAMES[:nchr]
        t0 = time.perf_counter()
        fa = os.path.join(root, "kmer.fa"); dats = []
        with open(fa, "wb") as f:
            for c in range(nchr):
                L = int(lens[c])
                f.write((">%s\n" % names[c]).encode()); bases[c][:L].cpu().numpy().tofile(f); f.write(b"\n")
        for c in range(nchr):
            L = int(lens[c])
            d = os.path.join(root, names[c] + ".dat"); dats.append(d)
            _write_dat(d, names[c], masks[c].cpu().numpy().view(np.uint8)[:(L + 7) // 8], hits[c][:L].cpu().numpy(), L)
        bam = os.path.join(root, "S.bam"); open(bam, "wb").write(b"")          # -b must exist even with -i (CanvasBin/Program.cs:148-153)
        ref = os.path.join(root, "WholeGenomeFasta"); os.mkdir(ref)
        t_inputs = time.perf_counter() - t0
        bdir = os.path.join(ROOT, "canvas_amd", "bin")
        env = dict(os.environ, CANVAS_TOOL_TIMING="1")

        REPS = 3

        def run_once(tool, argv):
            t_unix = time.time(); t = time.perf_counter()
            r = subprocess.run([os.path.join(bdir, tool)] + argv, capture_output=True, text=True, env=env)
            wall = time.perf_counter() - t; t_end_unix = time.time()
            ph = None
            for line in r.stderr.splitlines():
                if line.startswith('{"tool"'):
                    ph = json.loads(line)
            o = {"wall_seconds": round(wall, 3), "exit_code": r.returncode}
            if ph:
                # the tool's own phases lie between the first and the last statement of its main; what the caller's clock sees in front of and behind them — the loader and
                # the runtime's start in front, the kernel taking the process apart behind — are phases too: with them the phases add up to the wall time
                phases = {"startup": max(0.0, ph["main_entered_unix"] - t_unix)}
                phases.update(ph["phases"])
                phases["exit"] = max(0.0, t_end_unix - ph["leaving_unix"])
                o["phases"] = {k: round(v, 3) for k, v in phases.items()}
                o["unattributed_seconds"] = round(wall - sum(phases.values()), 3)
            if r.returncode != 0:
                o["stderr_tail"] = r.stderr[-300:]
            return o

        def run(tool, argv):
            """REPS runs of one tool (one OS process each, as CanvasRunner launches them): the MEDIAN run is reported with its phases, the minimum and every run's wall beside it"""
            runs = [run_once(tool, argv) for _ in range(REPS)]
            bad = [r for r in runs if r["exit_code"] != 0]
            if bad:
                return bad[0]
            order = sorted(range(REPS), key=lambda i: runs[i]["wall_seconds"])
            o = dict(runs[order[REPS // 2]])
            o["wall_seconds_min"] = runs[order[0]]["wall_seconds"]
            o["wall_seconds_of_each_run"] = [r["wall_seconds"] for r in runs]
            o["runs"] = REPS
            return o
        binned = os.path.join(root, "S.binned"); cleaned = os.path.join(root, "S.cleaned"); lsd = os.path.join(root, "LocalSD.txt")
        res = {"inputs": {"kmer_fa_GB": round(os.path.getsize(fa) / 1e9, 2), "dat_GB": round(sum(os.path.getsize(d) for d in dats) / 1e9, 2), "seconds_to_write_them": round(t_inputs, 1)}}
        argv = ["-b", bam, "-r", fa, "-o", binned, "-d", "100", "-m", "TruncatedDynamicRange"]
        for d in dats:
            argv += ["-i", d]
        if keep_dir:
            open(os.path.join(root, "bin_cmd.txt"), "w").write(" ".join([os.path.join(bdir, "CanvasBin")] + argv) + "\n")
            open(keep_dir, "w").write(root + "\n")
        res["CanvasBin"] = run("CanvasBin", argv)
        res["CanvasClean"] = run("CanvasClean", ["-i", binned, "-o", cleaned, "-g", "-s", "-r", "--local-sd-metric-file=" + lsd])
        for method in ("PerSampleHMM", "CBS", "Wavelets"):
            res["CanvasPartition -m " + method] = run("CanvasPartition", ["-i", cleaned, "-o", os.path.join(root, "S.%s.partitioned" % method), "-r", ref, "-m", method])
        ok = all(v.get("exit_code") == 0 for k, v in res.items() if k.startswith("Canvas"))
        res["all_exit_codes_zero"] = ok
        if ok:
            import gzip
            with gzip.open(cleaned, "rb") as f:
                res["cleaned_rows"] = sum(1 for _ in f)
            res["cleaned_rows_equal_the_library_call"] = bool(res["cleaned_rows"] == int(keep["n_out"]))
            wall = res["CanvasBin"]["wall_seconds"] + res["CanvasClean"]["wall_seconds"] + res["CanvasPartition -m PerSampleHMM"]["wall_seconds"]
            res["wall_seconds_bin_clean_partition(PerSampleHMM)"] = round(wall, 3)
            res["wall_seconds_bin_clean_partition(PerSampleHMM)_min"] = round(res["CanvasBin"]["wall_seconds_min"] + res["CanvasClean"]["wall_seconds_min"] + res["CanvasPartition -m PerSampleHMM"]["wall_seconds_min"], 3)
            res["wall_is"] = "sum of the three tools' MEDIAN walls over %d runs each (one OS process per run; `_min`: the sum of their fastest runs); phases of the median run, incl. startup (spawn -> main) and exit (main's last statement -> wait returns)" % REPS
            res["largest_unattributed_fraction"] = round(max(abs(v.get("unattributed_seconds", 0.0)) / v["wall_seconds"] for k, v in res.items() if k.startswith("Canvas") and isinstance(v, dict)), 3)
            res["bins_per_s_file_io_inclusive"] = round(int(keep["total"]) / wall, 1)
            if cpu and "seconds" in cpu:
                io = 0.0
                for k in ("CanvasBin", "CanvasClean", "CanvasPartition -m PerSampleHMM"):
                    ph = res[k].get("phases", {})
                    io += sum(v for n, v in ph.items() if n in ("read", "write", "startup", "exit"))
                est = io + float(cpu["seconds"]["total"])
                res["cpu_tools_estimate"] = {"seconds": round(est, 3), "file_io_seconds_of_the_same_tools": round(io, 3), "oracle_compute_seconds": cpu["seconds"]["total"], "threads": cpu.get("cores"),
                                             "note": "estimate: startup + read + write + exit phases of the drop-in tools (median runs) + the oracle's Bin + Clean + PerSampleHMM seconds (cpu_baseline)"}
                res["speedup_file_io_inclusive_vs_estimate"] = round(est / wall, 2)
        return res
    finally:
        if not keep_dir:
            shutil.rmtree(root, ignore_errors=True)


def somatic_flow(args, cv, torch, seed, bases, masks, lens, is_auto, flags, device):
    """BASELINE configs[4] at whole-genome size: tumour 80x (GCContentWeighted, fragment lengths, purity 0.7) + matched normal 40x over the same reference ->
    LSNorm ratio x 40 -> F2 -> CanvasClean -> F2 -> CBS (alpha 0.01, 10000 permutations).  Reported, not part of `value`; hand-off-by-hand-off parity with the
    chained oracle is tests/test_somatic_flow_gpu.py (sizes the oracle finishes in seconds); here the normal's bins of the two smallest autosomes are checked."""
    from canvas_amd import synth
    from canvas_amd.lib import synth_generate_sample_device
    rt, rn = args.rate * 4.0 / 3.0, args.rate * 2.0 / 3.0
    thr_t = torch.from_numpy(synth.poisson_thresholds(rt, purity=0.7).view(np.int32)).to(device)
    thr_n = torch.from_numpy(synth.poisson_thresholds(rn, flat=True).view(np.int32)).to(device)
    hits_t, fl_t, hits_n = [], [], []
    for c, L in enumerate(lens):
        h, f = synth_generate_sample_device(seed, seed + 1000, c, int(L), thr_t, device, with_fraglen=True); hits_t.append(h); fl_t.append(f)
        h, _ = synth_generate_sample_device(seed, seed + 2000, c, int(L), thr_n, device); hits_n.append(h)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = cv.tumor_normal_flow(bases, masks, hits_t, fl_t, hits_n, lens, is_auto, flags, 0.01, 10000)
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    r = cv.tumor_normal_flow(bases, masks, hits_t, fl_t, hits_n, lens, is_auto, flags, 0.01, 10000, keep=not args.no_cpu_baseline)
    sec = time.perf_counter() - t0
    o = {"seconds": round(sec, 3), "first_call_seconds": round(first, 3), "bins": int(r["n_bins"]), "bins_per_s": round(int(r["n_bins"]) / sec, 1), "bin_size": int(r["bin_size"]),
         "bins_with_ratio": int(r["n_ratio"]), "bins_after_clean": int(r["n_clean"]), "library_size_factor": r["library_size_factor"], "segments": int(r["segments"]),
         "cbs_tmaxo_calls": int(r["cbs_stats"][0]), "cbs_permutations": int(r["cbs_stats"][2]), "stage_seconds": r["stage_seconds"],
         "workload": "BASELINE configs[4]: tumour 80x (rate %.3f, purity 0.7, -m GCContentWeighted) / normal 40x (rate %.3f), LSNorm x 40, Clean, CBS" % (rt, rn)}
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        ok = True
        tchr = r["tumour"]["chr"].cpu().numpy(); tstop = r["tumour"]["stop"].cpu().numpy(); ncount = r["normal_count"].cpu().numpy()
        for c in (20, 21):
            L = int(lens[c])
            e = O.bin_chromosome(bases[c][:L].cpu().numpy(), masks[c].cpu().numpy().view(np.uint8), hits_n[c][:L].cpu().numpy(), int(r["bin_size"]), 3)
            sel = tchr == c
            ok &= bool((tstop[sel] == e[1]).all() and (ncount[sel] == e[3].astype(np.float32)).all())
        cov = r["cov"].cpu().numpy()
        o["normal_bins_chr21_22_vs_oracle"] = ok
        o["median_coverage"] = float(np.median(cov))                  # a diploid bin sits at ratio 1 x 40
        # the CBS stage at full size on the host cores (the oracle, one task per chromosome as CBSRunner.cs:62-89 does): seconds beside the GPU's, segments and RNG consumption compared
        cores = min(os.cpu_count() or 1, 24)
        off_h = r["chr_offset"]
        per = [np.ascontiguousarray(cov[off_h[c]:off_h[c + 1]]) for c in range(len(off_h) - 1)]
        t_o = time.perf_counter()
        exp_seg, est = O.cbs_genome(per, 0.01, 10000, threads=cores)
        o["cbs_oracle_seconds"] = round(time.perf_counter() - t_o, 3); o["cbs_oracle_threads"] = cores
        got = r["seg_len"].cpu().numpy(); nseg_c = r["nseg"]; cst = r["cbs_stats"]
        o["cbs_parity_vs_oracle"] = bool(all(int(nseg_c[c]) == len(exp_seg[c]) and (got[off_h[c]:off_h[c] + nseg_c[c]] == exp_seg[c]).all() for c in range(len(per)))
                                         and int(cst[0]) == int(est[0]) and int(cst[2]) == int(est[2]) and int(cst[4]) == int(est[4]))
    return o


def pedigree_flow(args, cv, torch, seed, bases, masks, lens, is_auto, flags, device):
    """BASELINE configs[3] at whole-genome size on one GPU: a trio over one reference (60x each) -> multi-sample bin size (median of the autosome rates of all samples,
    CanvasBin.cs:86-110) -> CanvasBin per sample on that size -> CanvasClean of the three samples in ONE cohort call (canvas_clean_batch) -> the bins every sample still has
    (MergeMultiSampleCleanedBedFile, Utilities.cs:834-920) -> F2 hand-off and PerSampleHMM per sample.  Reported, not part of `value`; the small-scale hand-off-by-hand-off
    parity incl. SplitOverlappingSegments is tests/test_pedigree_flow_gpu.py.  With the CPU baseline enabled: every sample's bins and cleaned bins and the merged list are
    compared with the oracle at full size."""
    from canvas_amd import synth
    from canvas_amd.lib import synth_generate_sample_device
    nchr = len(lens)
    thr = torch.from_numpy(synth.poisson_thresholds(args.rate).view(np.int32)).to(device)
    hits = [[synth_generate_sample_device(seed, seed + 3000 + 17 * s, c, int(L), thr, device)[0] for c, L in enumerate(lens)] for s in range(3)]
    torch.cuda.synchronize()
    cap = int(int(lens.sum()) // 100) + 16
    mk = lambda dt: torch.empty(cap, dtype=dt, device=device)
    outs = [dict(chr=mk(torch.int32), start=mk(torch.int32), stop=mk(torch.int32), gc=mk(torch.int32), count=mk(torch.float32)) for _ in range(3)]
    stage = {}

    def run(keep):
        t = [time.perf_counter()]

        def tick(name):
            cv.synchronize(); torch.cuda.synchronize()
            now = time.perf_counter(); stage[name] = round(now - t[0], 4); t[0] = now
        rates = []
        for s in range(3):
            _, _, r = cv.bin_rates(hits[s], masks, lens)
            rates += [r[c] for c in range(nchr) if is_auto[c]]
        bin_size = cv.bin_size_from_rates(rates, 100)
        tick("rates+bin_size")
        totals = []
        for s in range(3):
            _, per, total = cv.bin_genome(bases, masks, hits[s], lens, bin_size, 3, out=outs[s])
            totals.append(int(total))
        tick("bin x3")
        binned = [{k: v[:totals[s]].clone() for k, v in outs[s].items()} for s in range(3)] if keep else None
        t[0] = time.perf_counter()
        nout, lsd, _ = cv.clean_batch(outs, totals, is_auto, flags)
        tick("clean (cohort call)")
        mc, ms, me, mcnt, k = cv.merge_cleaned(outs, [int(x) for x in nout])
        off = cv.chromosome_offsets(mc, k, nchr)
        tick("merge")
        nseg = []
        for s in range(3):
            cov = cv.quantize_f2(mcnt[s], k)
            st = cv.hmm_per_sample(cov, off)
            change = (st[1:] != st[:-1])
            starts = torch.zeros(k, dtype=torch.bool, device=device); starts[torch.from_numpy(np.asarray(off[:-1][np.diff(off) > 0], np.int64)).to(device)] = True
            nseg.append(int((change | starts[1:]).sum().item()) + 1)
        tick("f2+hmm x3")
        return dict(bin_size=int(bin_size), totals=totals, nout=[int(x) for x in nout], lsd=[float(x) for x in lsd], merged=int(k), nseg=nseg, binned=binned,
                    merged_arrays=(mc, ms, me, mcnt) if keep else None)

    run(False)
    t0 = time.perf_counter()
    r = run(not args.no_cpu_baseline)
    sec = time.perf_counter() - t0
    o = {"seconds": round(sec, 3), "samples": 3, "bin_size": r["bin_size"], "bins_per_sample": r["totals"], "bins_after_clean": r["nout"], "bins_common_to_all": r["merged"],
         "segments_per_sample": r["nseg"], "stage_seconds": dict(stage), "bins_per_s": round(sum(r["totals"]) / sec, 1),
         "workload": "BASELINE configs[3]: trio, 60x each (rate %.3f), one reference; multi-sample bin size, CanvasBin x 3, CanvasClean as one cohort call, bin intersection, PerSampleHMM x 3" % args.rate}
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        cores = min(os.cpu_count() or 1, 24)
        hb = [b[:int(L)].cpu().numpy() for b, L in zip(bases, lens)]; hm = [m.cpu().numpy().view(np.uint8) for m in masks]
        is_y = np.zeros(nchr, np.uint8)
        ok_bin = ok_clean = True; exps = []
        t_o = time.perf_counter()
        for s in range(3):
            hh = [h[:int(L)].cpu().numpy() for h, L in zip(hits[s], lens)]
            res = O.bin_genome(hb, hm, hh, r["bin_size"], mode=3, threads=cores)
            e = dict(chr=np.concatenate([np.full(len(res[0][c]), c, np.int32) for c in range(nchr)]), start=np.concatenate(res[0]), stop=np.concatenate(res[1]),
                     gc=np.concatenate(res[2]), count=np.concatenate(res[3]).astype(np.float32))
            b = r["binned"][s]
            ok_bin &= bool(len(e["chr"]) == r["totals"][s] and (b["stop"].cpu().numpy() == e["stop"]).all() and (b["count"].cpu().numpy() == e["count"]).all() and (b["gc"].cpu().numpy() == e["gc"]).all())
            ex = O.clean(e["chr"], e["start"], e["stop"], e["count"], e["gc"], is_auto, is_y, flags)
            n = r["nout"][s]
            ok_clean &= bool(n == len(ex["chr"]) and ex["local_sd"] == r["lsd"][s] and (outs[s]["count"][:n].cpu().numpy().view(np.uint32) == ex["count"].view(np.uint32)).all()
                             and (outs[s]["start"][:n].cpu().numpy() == ex["start"]).all())
            exps.append(ex)
        ec, es, ee, ecnt = O.merge_cleaned(exps)
        mc, ms, me, mcnt = r["merged_arrays"]
        ok_merge = bool(len(ec) == r["merged"] and (ms.cpu().numpy() == es).all() and (me.cpu().numpy() == ee).all()
                        and all((mcnt[s].cpu().numpy().view(np.uint32) == ecnt[s].view(np.uint32)).all() for s in range(3)))
        o["oracle_seconds_bin_clean_merge"] = round(time.perf_counter() - t_o, 3); o["oracle_threads"] = cores
        o["parity_vs_oracle"] = {"bins_x3": ok_bin, "clean_x3_bitexact": ok_clean, "merged_bins": ok_merge}
    return o


def cpu_baseline(keep, host, lens, is_auto, flags, total_bases):
    """The CPU oracle (oracle/, a restatement of the reference's algorithm — the C# original cannot be built here) timed on this box's cores on the WHOLE
    workload (no extrapolation), and used at the same time as a full-size parity check of the GPU result.  Threads as the reference uses them: CanvasBin and
    CanvasPartition one task per chromosome (Parallel.ForEach, CanvasBin.cs:513-539, HiddenMarkovModelsRunner.cs:51-58), CanvasClean single-threaded."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    cores = os.cpu_count() or 1
    par = min(cores, len(lens))
    hb = [host["bases"][c].numpy()[:int(lens[c])] for c in range(len(lens))]
    hh = [host["hits"][c].numpy()[:int(lens[c])] for c in range(len(lens))]
    hm = [host["masks"][c].numpy().view(np.uint8) for c in range(len(lens))]
    t0 = time.perf_counter()
    rates = O.bin_rates_genome(hm, hh, threads=par)
    bs = O.bin_size([r for r, a in zip(rates, is_auto) if a], 100)
    res = O.bin_genome(hb, hm, hh, bs, 3, threads=par)
    t_bin = time.perf_counter() - t0
    binned = {k: v.cpu().numpy() for k, v in keep["binned"].items()}
    ok_bins = bool(bs == keep["bin_size"] and (binned["stop"] == np.concatenate(res[1])).all() and (binned["count"] == np.concatenate(res[3]).astype(np.float32)).all()
                   and (binned["gc"] == np.concatenate(res[2])).all() and (binned["start"] == np.concatenate(res[0])).all())
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
    return {"value": round(keep["total"] / t_total, 1), "unit": "bins/s", "cores": par, "host_cores": cores, "kind": "port",
            "sample": f"the whole workload, no extrapolation: CanvasBin (rates + bins) on all {total_bases} bases with {par} threads (one task per chromosome), "
                      f"CanvasClean (1 thread) and PerSampleHMM ({par} threads) on all {keep['total']} bins",
            "seconds": {"bin": round(t_bin, 3), "clean": round(t_clean, 3), "hmm": round(t_hmm, 3), "total": round(t_total, 3)},
            "parity_vs_gpu": {"bins_whole_genome": ok_bins, "clean_bitexact": ok_clean, "viterbi_states": ok_states},
            "note": "C++ restatement of the C# reference (which cannot be built here), -O2; a reported baseline, not a target"}


def sharded_main(args, cv, rank, world, device):
    from canvas_amd import parallel
    return parallel.bench_sharded(args, cv, rank, world, device, HBM_PEAK_GBS)


if __name__ == "__main__":
    main()
