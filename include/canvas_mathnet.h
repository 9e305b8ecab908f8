/* canvas_mathnet.h — the ONE assumption about MathNet.Numerics 3.17 that nobody could check where this code was written.
 *
 * CanvasPartition -m CBS seeds one generator per chromosome from a master generator (/root/reference/Src/Canvas/CanvasPartition/CBSRunner.cs:107-112):
 *
 *     var seedGenerator = new MersenneTwister(0);
 *     perChromosomeRandom[chr] = new MersenneTwister(seedGenerator.NextFullRangeInt32(), true);
 *
 * MathNet is a NuGet dependency that is not under /root/reference, and no .NET SDK exists in the build image.  NextFullRangeInt32() is
 * BitConverter.ToInt32 of four bytes taken from the generator (RandomSource.NextBytes -> DoSampleBytes); which eight bits of a 32-bit output make a
 * byte is the open question.  Every per-chromosome seed — and through it every permutation decision of CBS — depends on the answer, and the product and
 * the CPU oracle share it, so no test in this repository can see a wrong guess.  Both read the variant from HERE and nowhere else:
 *
 *     0  byte = (byte)(genrand_int32() % 256)            MersenneTwister's own DoSampleBytes override (the working assumption)
 *     1  byte = (byte)((genrand_int32() >> 1) % 256)     RandomSource.DoSampleBytes over DoSampleInteger() = (int)(genrand_int32() >> 1)
 *     2  byte = (byte)(genrand_int32() >> 24)            RandomSource.DoSampleBytes over (int)(NextDouble() * 256)
 *
 * To settle it on a machine with a .NET SDK:   new MersenneTwister(0).NextFullRangeInt32()   prints
 *     variant 0:  -1066061908   (bytes AC 2F 75 C0)      variant 1:  1614419798   (56 17 3A 60)      variant 2:  -659056756   (8C 97 B7 D8)
 * (the first four outputs of init_genrand(0) are 8C7F0AAC 97C4AA2F B716A675 D821CCC0).  tests/test_mathnet_seed_variants.py derives all three from numpy's
 * MT19937 (same init_genrand) and checks the product (canvas_cbs_seeds) and the oracle (orc_cbs_seeds) against them.
 *
 * Selection: compile-time default CANVAS_MATHNET_SEED_BYTES below; the environment variable CANVAS_MATHNET_SEED_BYTES=0|1|2 overrides it at run time in
 * the product AND the oracle (read once per process), so flipping it needs no rebuild. */
#ifndef CANVAS_MATHNET_H
#define CANVAS_MATHNET_H
#include <stdint.h>
#include <stdlib.h>

#ifndef CANVAS_MATHNET_SEED_BYTES
#define CANVAS_MATHNET_SEED_BYTES 0
#endif

/* the variant in force in this process */
static inline int canvas_mathnet_seed_variant(void) {
    const char* e = getenv("CANVAS_MATHNET_SEED_BYTES");
    if (e && e[0] >= '0' && e[0] <= '2' && e[1] == 0) return e[0] - '0';
    return CANVAS_MATHNET_SEED_BYTES;
}
/* one byte of NextBytes() from one 32-bit output of genrand_int32() */
static inline uint32_t canvas_mathnet_seed_byte(uint32_t genrand, int variant) {
    return variant == 1 ? ((genrand >> 1) & 0xFFu) : variant == 2 ? (genrand >> 24) : (genrand & 0xFFu);
}
/* NextFullRangeInt32() from four consecutive outputs, little-endian */
static inline int32_t canvas_mathnet_full_range_int32(const uint32_t genrand4[4], int variant) {
    uint32_t v = 0;
    for (int b = 0; b < 4; b++) v |= canvas_mathnet_seed_byte(genrand4[b], variant) << (8 * b);
    return (int32_t)v;
}
#endif
