// CanvasPartition with the GPU library: the patch to CanvasPartition.Main (Src/Canvas/CanvasPartition/CanvasPartition.cs:24-190).
// Kept from the module: the OptionSet (:42-57) with every option CanvasRunner passes (-p always, --evenness-metric-file for Somatic-WGS, -c for
// pedigrees; CanvasRunner.cs:904-971), the exit conventions (:60-100), SegmentationInput (file reading, GenomicBinFilter), DeriveSegments,
// SplitOverlappingSegments, PostProcessSegments and WriteCanvasPartitionResults.  Replaced: the four runners' compute.
// NOT COMPILED HERE (no dotnet SDK in the image); canvas_amd/tools/canvas_partition_main.cpp is the same program in C++ and is what the tests run.
using System;
using System.Collections.Generic;
using System.Linq;
using CanvasCommon;
using static CanvasHipInterop.CanvasHip;

namespace CanvasPartition
{
    /// <summary>Coverage of one SegmentationInput laid out for the library: chromosomes in CoverageInfo order, concatenated.</summary>
    sealed class HipCoverage : IDisposable
    {
        public readonly List<string> Chromosomes; public readonly long[] ChrOffset; public readonly DeviceBuffer Cov; readonly IntPtr _ctx;
        public HipCoverage(IntPtr ctx, CoverageInfo info)
        {
            _ctx = ctx; Chromosomes = info.CoverageByChr.Keys.ToList();
            ChrOffset = new long[Chromosomes.Count + 1];
            for (int c = 0; c < Chromosomes.Count; c++) ChrOffset[c + 1] = ChrOffset[c] + info.CoverageByChr[Chromosomes[c]].Length;
            var all = new double[ChrOffset[Chromosomes.Count]];
            for (int c = 0; c < Chromosomes.Count; c++) info.CoverageByChr[Chromosomes[c]].CopyTo(all, ChrOffset[c]);
            Cov = new DeviceBuffer(ctx, 8L * all.Length);
            Check(ctx, canvas_memcpy_h2d(ctx, Cov.Ptr, all, 8L * all.Length), "upload coverage");
        }
        public void Dispose() { Cov.Dispose(); }
    }

    static class HipPartition
    {
        /// <summary>WaveletsRunner.Run (WaveletsRunner.cs:52-81): evenness metric, breakpoints per chromosome, DeriveSegments for the chromosomes of VafByChr.</summary>
        public static Dictionary<string, SegmentationInput.Segment[]> Wavelets(IntPtr ctx, SegmentationInput input, bool isGermline, CanvasPartitionParameters p)
        {
            using (var cov = new HipCoverage(ctx, input.CoverageInfo))
            {
                int nchr = cov.Chromosomes.Count;
                if (!string.IsNullOrEmpty(input.EvennessMetricFile))
                {
                    Check(ctx, canvas_evenness_score(ctx, nchr, cov.Cov.Ptr, cov.ChrOffset, p.EvennessScoreWindow, out double score, out int valid), "canvas_evenness_score");
                    if (valid != 0) CanvasIO.WriteEvennessMetricToTextFile(input.EvennessMetricFile, score);          // IO.cs:88-98
                    else Console.Error.WriteLine("Unable to calculate an evenness score, using coverage for segmentation");
                }
                long cap = cov.ChrOffset[nchr] + nchr + 1; var bps = new int[cap]; var bpOffset = new long[nchr + 1];
                Check(ctx, canvas_wavelets(ctx, nchr, cov.Cov.Ptr, cov.ChrOffset, isGermline ? 1 : 0, p.ThresholdLowerMaf, 80.0, p.MadFactor, p.EvennessScoreWindow, 10, bps, cap, bpOffset),
                      "canvas_wavelets");
                var segments = new Dictionary<string, SegmentationInput.Segment[]>();
                for (int c = 0; c < nchr; c++)
                {
                    string chr = cov.Chromosomes[c];
                    if (!input.VafByChr.ContainsKey(chr)) continue;                                                   // WaveletsRunner.cs:75
                    var list = new List<int>(); for (long k = bpOffset[c]; k < bpOffset[c + 1]; k++) list.Add(bps[k]);
                    segments[chr] = SegmentationInput.DeriveSegments(list, input.CoverageInfo.CoverageByChr[chr].Length, input.CoverageInfo.StartByChr[chr], input.CoverageInfo.EndByChr[chr]);
                }
                return segments;
            }
        }

        /// <summary>CBSRunner.Run (CBSRunner.cs:40-151).</summary>
        public static Dictionary<string, SegmentationInput.Segment[]> Cbs(IntPtr ctx, SegmentationInput input, SegmentSplitUndo undo, double alpha)
        {
            using (var cov = new HipCoverage(ctx, input.CoverageInfo))
            {
                int nchr = cov.Chromosomes.Count; long n = cov.ChrOffset[nchr];
                using (var dLen = new DeviceBuffer(ctx, 4L * (n + 1)))
                {
                    var nseg = new int[nchr]; var stats = new long[8];
                    int undoCode = undo == SegmentSplitUndo.Prune ? 1 : undo == SegmentSplitUndo.SDUndo ? 2 : 0;
                    Check(ctx, canvas_cbs_undo(ctx, nchr, cov.Cov.Ptr, cov.ChrOffset, alpha, 10000, undoCode, 3.0, dLen.Ptr, nseg, stats), "canvas_cbs_undo");
                    var lens = new int[n + 1]; Check(ctx, canvas_memcpy_d2h(ctx, lens, dLen.Ptr, 4L * (n + 1)), "download");
                    var result = new Dictionary<string, SegmentationInput.Segment[]>();
                    for (int c = 0; c < nchr; c++)
                    {
                        string chr = cov.Chromosomes[c]; uint[] s = input.CoverageInfo.StartByChr[chr], e = input.CoverageInfo.EndByChr[chr];
                        var segs = new SegmentationInput.Segment[nseg[c]]; long first = 0;
                        for (int k = 0; k < nseg[c]; k++) { long len = lens[cov.ChrOffset[c] + k]; segs[k] = new SegmentationInput.Segment { start = s[first], end = e[first + len - 1] }; first += len; }   // CBSRunner.cs:127-137
                        result[chr] = segs;
                    }
                    return result;
                }
            }
        }

        /// <summary>HiddenMarkovModelsRunner.Run (HiddenMarkovModelsRunner.cs:23-109): per sample (one input) or joint (all inputs share the bins).</summary>
        public static Dictionary<string, SegmentationInput.Segment[]> Hmm(IntPtr ctx, List<SegmentationInput> inputs, bool isPerSample)
        {
            var covs = inputs.Select(i => new HipCoverage(ctx, i.CoverageInfo)).ToList();
            try
            {
                var first = covs[0]; int nchr = first.Chromosomes.Count; long n = first.ChrOffset[nchr];
                using (var dState = new DeviceBuffer(ctx, 4L * n))
                {
                    if (isPerSample) Check(ctx, canvas_hmm_per_sample(ctx, nchr, first.Cov.Ptr, first.ChrOffset, dState.Ptr), "canvas_hmm_per_sample");
                    else Check(ctx, canvas_hmm_joint(ctx, covs.Count, nchr, covs.Select(c => c.Cov.Ptr).ToArray(), first.ChrOffset, dState.Ptr), "canvas_hmm_joint");
                    var state = new int[n]; Check(ctx, canvas_memcpy_d2h(ctx, state, dState.Ptr, 4L * n), "download");
                    var result = new Dictionary<string, SegmentationInput.Segment[]>();
                    var info = inputs[0].CoverageInfo;
                    for (int c = 0; c < nchr; c++)
                    {
                        string chr = first.Chromosomes[c]; long b0 = first.ChrOffset[c]; int len = (int)(first.ChrOffset[c + 1] - b0);
                        if (len <= 10) continue;                                                                       // :69, state -1 in the library
                        var bp = new List<int> { 0 };
                        for (int i = 1; i < len; i++) if (state[b0 + i] != state[b0 + i - 1]) bp.Add(i);               // :88-95
                        result[chr] = SegmentationInput.DeriveSegments(bp, len, info.StartByChr[chr], info.EndByChr[chr]);
                    }
                    return result;
                }
            }
            finally { foreach (var c in covs) c.Dispose(); }
        }
        // In Main's switch (:116-183) the four `new XRunner(...).Run(...)` calls become HipPartition.Wavelets / Cbs / Hmm with one context
        // (`IntPtr ctx = canvas_create(0)`, failure -> message + return 1); `referencePloidy`, SplitOverlappingSegments and PostProcessAndWriteResults
        // (:114, :138-143, :185-189) stay as they are.  With `-m CBS` the context is created BEFORE the inputs are read (:102-112) and `canvas_cbs_prefetch(ctx, 25, 16 << 20)`
        // follows it at once: the per-chromosome MersenneTwister streams (CBSRunner.cs:107-112) are constants, and the library generates them on its own thread while
        // CanvasSegment.ReadBedInput parses the .cleaned files.  A host that keeps the bins on the device can let the library number the segments as well:
        // canvas_segment_ids_ploidy takes the -b intervals and the -p records and returns the id column of the .partitioned file.
    }
}
