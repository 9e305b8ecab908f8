// CanvasClean with the GPU library: the patch to CanvasClean.Main (Src/Canvas/CanvasClean/CanvasClean.cs:415-533).
// Everything before the compute stays as it is in the module — option parsing (:431-446), the help / missing-file exits (:455-472),
// CanvasIO.ReadFromTextFile — and so does the writing at the end.  RemoveBigBins .. RemoveBinsWithExtremeLocalSD (:475-530) become ONE call.
// NOT COMPILED HERE (no dotnet SDK in the image); canvas_amd/tools/canvas_clean_main.cpp is the same program in C++ and is what the tests run.
using System;
using System.Collections.Generic;
using System.Linq;
using CanvasCommon;
using Isas.SequencingFiles;
using static CanvasHipInterop.CanvasHip;

namespace CanvasClean
{
    static class HipClean
    {
        /// <returns>the cleaned bins; localSd < 0 when the metric does not apply (fewer than 50000 bins or no --local-sd-metric-file)</returns>
        public static List<SampleGenomicBin> Run(List<SampleGenomicBin> bins, bool doGCnorm, bool doSizeFilter, bool doOutlierRemoval, bool wantLocalSd,
            CanvasGCNormalizationMode mode, int minBinsPerGCForWeightedMedian, out double localSd)
        {
            // chromosome index = order of first appearance in the file (what the C# dictionaries enumerate)
            var chromIndex = new Dictionary<string, int>(); var chromNames = new List<string>();
            foreach (var b in bins) if (!chromIndex.ContainsKey(b.GenomicBin.Chromosome)) { chromIndex[b.GenomicBin.Chromosome] = chromNames.Count; chromNames.Add(b.GenomicBin.Chromosome); }
            int n = bins.Count, nchr = Math.Max(1, chromNames.Count);
            int[] chr = new int[n], start = new int[n], stop = new int[n], gc = new int[n]; float[] count = new float[n];
            for (int i = 0; i < n; i++) { var b = bins[i]; chr[i] = chromIndex[b.GenomicBin.Chromosome]; start[i] = b.Start; stop[i] = b.Stop; gc[i] = b.GenomicBin.GC; count[i] = b.Count; }
            byte[] isAutosome = chromNames.Select(c => (byte)(GenomeMetadata.SequenceMetadata.IsAutosome(c) ? 1 : 0)).DefaultIfEmpty((byte)0).ToArray();
            byte[] isY = chromNames.Select(c => (byte)(c == "chrY" || c == "Y" ? 1 : 0)).DefaultIfEmpty((byte)0).ToArray();      // LoessGCNormalizer.cs:49-50
            uint flags = (doGCnorm ? CleanGcNorm : 0) | (doSizeFilter ? CleanFiltSize : 0) | (doOutlierRemoval ? CleanOutliers : 0) | (wantLocalSd ? CleanLocalSd : 0)
                       | (mode == CanvasGCNormalizationMode.LOESS ? CleanLoess : 0);
            IntPtr ctx = canvas_create(0);
            if (ctx == IntPtr.Zero) throw new InvalidOperationException("no usable GPU (libcanvas_hip has no CPU fallback)");
            try
            {
                long bytes = 4L * n;
                using (var dChr = new DeviceBuffer(ctx, bytes)) using (var dStart = new DeviceBuffer(ctx, bytes)) using (var dStop = new DeviceBuffer(ctx, bytes))
                using (var dGc = new DeviceBuffer(ctx, bytes)) using (var dCount = new DeviceBuffer(ctx, bytes))
                {
                    Check(ctx, canvas_memcpy_h2d(ctx, dChr.Ptr, chr, bytes), "upload"); Check(ctx, canvas_memcpy_h2d(ctx, dStart.Ptr, start, bytes), "upload");
                    Check(ctx, canvas_memcpy_h2d(ctx, dStop.Ptr, stop, bytes), "upload"); Check(ctx, canvas_memcpy_h2d(ctx, dGc.Ptr, gc, bytes), "upload");
                    Check(ctx, canvas_memcpy_h2d(ctx, dCount.Ptr, count, bytes), "upload");
                    var info = new int[8];
                    Check(ctx, canvas_clean2(ctx, n, dChr.Ptr, dStart.Ptr, dStop.Ptr, dCount.Ptr, dGc.Ptr, nchr, isAutosome, isY, flags, minBinsPerGCForWeightedMedian,
                                             out localSd, out long nOut, info), "canvas_clean2");
                    long ob = 4L * nOut;
                    Check(ctx, canvas_memcpy_d2h(ctx, chr, dChr.Ptr, ob), "download"); Check(ctx, canvas_memcpy_d2h(ctx, start, dStart.Ptr, ob), "download");
                    Check(ctx, canvas_memcpy_d2h(ctx, stop, dStop.Ptr, ob), "download"); Check(ctx, canvas_memcpy_d2h(ctx, gc, dGc.Ptr, ob), "download");
                    Check(ctx, canvas_memcpy_d2h(ctx, count, dCount.Ptr, ob), "download");
                    var cleaned = new List<SampleGenomicBin>((int)nOut);
                    for (int i = 0; i < nOut; i++) cleaned.Add(new SampleGenomicBin(chromNames[chr[i]], start[i], stop[i], gc[i], count[i]));
                    return cleaned;
                }
            }
            finally { canvas_destroy(ctx); }
        }
        // In Main, after `bins = CanvasIO.ReadFromTextFile(inFile)` (:475):
        //     var cleaned = HipClean.Run(bins, doGCnorm, doSizeFilter, doOutlierRemoval, localSdMetricFile != null, gcNormalizationMode, minNumberOfBinsPerGCForWeightedMedian, out double localSd);
        //     if (localSdMetricFile != null && localSd >= 0) CanvasIO.WriteLocalSdMetricToTextFile(localSdMetricFile, localSd);   // IO.cs:83-86
        //     CanvasIO.WriteToTextFile(outFile, cleaned);                                                                            // :532
        // A manifest (-t) keeps the module's own C# path: the library returns CANVAS_ERR_UNSUPPORTED for it.
    }
}
