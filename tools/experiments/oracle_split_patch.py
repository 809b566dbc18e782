#!/usr/bin/env python3
"""Round 4, the one bounded kernel experiment: how much could ANY histogram-free (splitter-based) selection gain?

Builds tools/kb/kb_ship (the shipped k_row_stats) and tools/kb/kb_oracle: a copy of the kernel in which the histogram,
the tile-0 range estimate and the whole `locate` step are gone and the two splitters come FOR FREE from an oracle -- a
per-row (first key, log2 width) pair computed on the host from the very data, bounding a bin of <= 48 members around the
median.  What is left is what every splitter scheme must still do: count the keys below the lower splitter (v_cmp +
s_bcnt1 per key, zero LDS atomics), ONE cross-wave exchange of the counts, collect the bin's members, rank them in one
wave.  A real scheme (Floyd-Rivest bounds from a sample) has to FIND its splitters and ends up with a window of a few
hundred keys (O(sqrt n)), so kb_oracle is an upper bound on its gain.  Results: tools/experiments/README.md (round 4).

    python tools/experiments/oracle_split_patch.py     # writes tools/kb/src_oracle.hip, builds both binaries
"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(REPO, "nvidia-resiliency-ext_amd", "csrc", "nvrx_straggler.hip")
OUT = os.path.join(REPO, "tools", "kb")


def sub(text, old, new, count=1):
    assert text.count(old) >= count, (old[:80], text.count(old))
    return text.replace(old, new, count) if count == 1 else text.replace(old, new)


def main():
    os.makedirs(OUT, exist_ok=True)
    # the experiment is pinned to the kernel text it was measured against (the shipped kernel of rounds 2-3, before the range
    # hint that came out of it was added)
    s = subprocess.check_output(["git", "-C", REPO, "show", "c89d9f0:nvidia-resiliency-ext_amd/csrc/nvrx_straggler.hip"], text=True)
    open(os.path.join(OUT, "src_ship_r03.hip"), "w").write(s)
    # the oracle: per launched row {first key of the median's bin, log2 of the bin width}
    s = sub(s, "template <int THREADS, int VPT>\n__global__ __launch_bounds__(THREADS) void k_row_stats(",
            "__device__ uint32_t g_oracle[8192][2];\n\ntemplate <int THREADS, int VPT>\n__global__ __launch_bounds__(THREADS) void k_row_stats(")
    s = sub(s, "        float4 x[VPT];\n", "        const uint32_t obase = g_oracle[blockIdx.x][0], osh = g_oracle[blockIdx.x][1];\n        uint32_t wbelow = 0u;\n        float4 x[VPT];\n")
    # tile 0: keys only -- no wave reduction, no LDS atomics, no barrier (1), no range estimate
    s = sub(s, """            uint32_t a = kmn, b = kmx;
            wave_minmax_u32(a, b);
            if (lane == 0) {
                atomicMin(&s_mm[0], a);
                atomicMax(&s_mm[1], b);
            }
        }
        __syncthreads();  // (1) tile-0 range
""", """        }
""")
    s = sub(s, """            const uint32_t mn0 = uni(s_mm[0]), mx0 = uni(s_mm[1]);
            const uint32_t R = speculative ? mx0 - mn0 : 0u;
            lo0 = mn0 > R ? mn0 - R : 0u;
            const uint32_t hi0 = mx0 < 0xFFFFFFFFu - R ? mx0 + R : 0xFFFFFFFFu;
            const int top = 32 - __clz((int)(hi0 - lo0));  // (hi0 - lo0) >> sh < 2048
            sh = (uint32_t)(top > HIST_BITS ? top - HIST_BITS : 0);
""", """            lo0 = obase;
            sh = osh;
""")
    # the streaming pass: a count of the keys below the lower splitter instead of one LDS atomic per key
    s = sub(s, "                    if (NVRX_ABLATE == 0) atomicAdd(&s_hist[min(__builtin_elementwise_sub_sat(kk, lo0) >> sh, (uint32_t)(HIST_BINS - 1))], 1u);",
            "                    wbelow += (uint32_t)__popcll(__ballot(kk < obase));")
    s = sub(s, """                    if (NVRX_ABLATE == 0 && valid)
                        atomicAdd(&s_hist[min(__builtin_elementwise_sub_sat(kk, lo0) >> sh, (uint32_t)(HIST_BINS - 1))], 1u);""",
            "                    wbelow += (uint32_t)__popcll(__ballot(valid && kk < obase));")
    # the one exchange: per-wave counts travel with the partial sums of barrier (2)
    s = sub(s, "            s_d[wave] = sum;\n        }\n        NVRX_PHASE(2);", "            s_d[wave] = sum;\n            s_sum[wave] = wbelow;\n        }\n        NVRX_PHASE(2);")
    # no locate: the bin is the oracle's, the rank inside it follows from the count
    s = sub(s, """            if (!conv) {
                bin = locate(k, pop);
                rebuild = speculative && (bin == 0u || bin == (uint32_t)(HIST_BINS - 1));
            }""", """            if (!conv) {
                uint32_t below = 0u;
#pragma unroll
                for (int w = 0; w < WAVES; w++) below += s_sum[w];
                k = k_rank - below;
                pop = 48u;
                bin = 0u;
                rebuild = false;
            }""")
    open(os.path.join(OUT, "src_oracle.hip"), "w").write(s)
    hipcc = "/opt/rocm/bin/hipcc"
    common = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{REPO}/include", "-Wno-unused-function", "-Wno-unused-value",
              "-Wno-unused-variable", "-Wno-unused-but-set-variable"]
    subprocess.check_call(common + [f'-DNVRX_SRC="{OUT}/src_ship_r03.hip"', os.path.join(REPO, "tools", "kbench.cpp"), "-o", os.path.join(OUT, "kb_ship")])
    subprocess.check_call(common + [os.path.join(REPO, "tools", "kbench.cpp"), "-o", os.path.join(OUT, "kb_head")])  # the tree's kernel (KB_HINT=1: with range hints)
    subprocess.check_call(common + ["-DNVRX_ORACLE_SPLIT", f'-DNVRX_SRC="{OUT}/src_oracle.hip"', os.path.join(REPO, "tools", "kbench.cpp"),
                                    "-o", os.path.join(OUT, "kb_oracle")])
    print("built", os.path.join(OUT, "kb_ship"), os.path.join(OUT, "kb_oracle"))


if __name__ == "__main__":
    main()
