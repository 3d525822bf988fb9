"""tools/phase_timeline.py — where does a wave of the evaluation sweep spend its life?  (GPU box; diagnostic, not product)

The C4 evaluation sweep (double pendulum 80^4) keeps three units 50-70 % busy — VALU issue, the vector-memory path, HBM —
and a substitution test says their times ADD (profiles/r03/negative_results.txt (9)).  This tool records, per wave, the
shader-clock time (s_memtime) at the phase boundaries of each chunk and folds the records of the waves that shared a SIMD
into a timeline: how many of a SIMD's wave slots are occupied over time and which phase every resident wave is in.

The traced kernel is built HERE, at run time, from the product's own translation unit (`pi_kernel_source`) plus the text
below — a copy of pi_eval_sweep_kernel's body (same helpers, same arithmetic, same launch geometry, <= 64 VGPRs and <= 80
SGPRs checked from the compiler's resource report) with stamps written into lanes of ONE VGPR by v_writelane and stored
once per wave at its end (256 B), by the waves of CU 0 of every shader array only.  Nothing of it is part of the library.

Variants (same arithmetic; V' must equal the product kernel's bit for bit — checked):
  product      the library's kernel through the C ABI (reference time)
  copy         the copy below without stamps (does the copy itself run like the product?)
  traced       the copy with stamps -> timeline
  persistent   `--persistent W`: W workgroups per XCD, each walking chunks c0 + j, c0 + j + W, ... of its XCD's slab with no
               workgroup boundary in between (no table re-staging, no barrier, no dispatch gap, no drain tail)

usage: python tools/phase_timeline.py [--env E] [--bins B] [--out profiles/r04/phase_timeline_c4.txt] [--persistent 64,128]
"""
import argparse
import ctypes
import os
import re
import json
import subprocess
import sys
import tempfile
from collections import defaultdict
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

from dynamicprogramming_amd import envs

PT_KERNELS = r'''
// ---- diagnostic copies of pi_eval_sweep_kernel (tools/phase_timeline.py) ----
#ifndef PT_TRACE
#define PT_TRACE 0
#endif
#ifndef PT_REORDER
#define PT_REORDER 0
#endif
#ifndef PT_PF_DIST
#define PT_PF_DIST 0
#endif
#ifndef PT_AFFINE
#define PT_AFFINE 0
#endif
#ifndef PT_LDSBOX
#define PT_LDSBOX 0
#endif
// PT_WPRIO: issue priority by position in the workgroup.  The arbiter prefers the oldest wave, so the 16 waves of a
// workgroup drift apart in index order and the first ones leave their slots empty for thousands of cycles before the
// last one frees the workgroup's resources.  1: base priority = wave / 4 (0..3), 3 while the gather is in flight;
// 2: base = wave / 8 (0..1), 2 while the gather is in flight; 3: base = wave / 4, no gather window.
#ifndef PT_WPRIO
#define PT_WPRIO 0
#endif
__device__ __forceinline__ void pt_setprio(int level) {
    switch (level) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
}
#if PT_AFFINE
// Grid coordinates without the LDS copy of the bin tables: x_d = (float)((double)i_d * step_d + start_d), two float64
// operations and one conversion, verified on the host to reproduce EVERY float32 entry of the table (the tables are
// np.linspace outputs, i.e. float64 affine sequences rounded to float32).  The action value of a lane comes out of a
// register that holds the action table across the lanes of the wave (ds_bpermute: the LDS crossbar, no LDS memory).
// Nothing is staged, so the workgroup needs no barrier and a wave's first dependence on memory is its policy entry.
constexpr double PT_AFF_A[PI_D] = PT_AFF_A_INIT;
constexpr double PT_AFF_B[PI_D] = PT_AFF_B_INIT;
__device__ __forceinline__ void pt_state_coords(unsigned int s, float (&x)[PI_D]) {
    unsigned int r = s;
#pragma unroll
    for (int d = PI_D - 1; d > 0; --d) {
        unsigned int q = r / (unsigned int)PI_GRID.g[d];
        x[d] = (float)((double)(r - q * (unsigned int)PI_GRID.g[d]) * PT_AFF_B[d] + PT_AFF_A[d]);
        r = q;
    }
    x[0] = (float)((double)r * PT_AFF_B[0] + PT_AFF_A[0]);
}
#endif
#define PT_WORDS 64
#if PT_TRACE
#define PT_PUT(value, slot) asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(tr) : "s"((int)(value)), "n"(slot))
#define PT_STAMP(slot) do { __builtin_amdgcn_sched_barrier(0);                                              \
        PT_PUT((unsigned int)__builtin_amdgcn_s_memtime(), slot);                                           \
        __builtin_amdgcn_sched_barrier(0); } while (0)
#define PT_STAMP2(p) do { if (k == 0) PT_STAMP(4 + (p)); else PT_STAMP(10 + (p)); } while (0)
#define PT_WAIT_ALL() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#define PT_WAIT_BUT1() asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory")
#else
#define PT_STAMP(slot) do {} while (0)
#define PT_STAMP2(p) do {} while (0)
#define PT_WAIT_ALL() do {} while (0)
#define PT_WAIT_BUT1() do {} while (0)
#endif

// PT_PERSISTENT == 0: the product's schedule (pi_first_chunk: cpw consecutive chunks per workgroup, workgroups dealt to
// XCDs in contiguous runs).  1: gridDim.x / 8 workgroups per XCD walk the XCD's slab of chunks interleaved.
// Traced iterations: it0 and it0 + 1 of the workgroup's chunk loop (slots 4 + 6 k .. 9 + 6 k).
template <int PERSISTENT>
__device__ __forceinline__ void pt_eval_body(const float* __restrict__ V, float* __restrict__ Vn,
                     const int* __restrict__ policy, const float* __restrict__ tab, long long s_begin, long long s_end,
                     float gamma, int cpw, unsigned int* __restrict__ trace, unsigned int* __restrict__ trace_count,
                     unsigned int trace_cap, int it0) {
    __shared__ float lds_tab[PI_GRID.tab_len];
    int tr = 0;
    PiStateIn nxt_init;
    nxt_init.v_old = 0.0f; nxt_init.action = 0; nxt_init.term = 0;
    PT_STAMP(2);
#if PT_TRACE
    PT_WAIT_ALL();                                      // the kernel arguments (s_load from the kernarg segment) are in
    PT_STAMP(21);
#endif
    const unsigned int tid = threadIdx.x;
#if PT_WPRIO
    const int wave_in_wg = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const int waves = PI_BLOCK_EVAL / 64;
    const int base_prio = PT_WPRIO == 2 ? (wave_in_wg * 2) / waves : (wave_in_wg * 4) / waves;
    pt_setprio(base_prio);
#endif
    long long c, c_step;
    long long c_end;
    if (PERSISTENT) {
        const long long n_chunks = (s_end - s_begin + PI_BLOCK_EVAL - 1) / PI_BLOCK_EVAL;
        const long long span = (n_chunks + PI_NXCD - 1) / PI_NXCD;
        const long long x = blockIdx.x % PI_NXCD, j = blockIdx.x / PI_NXCD;
        c_step = gridDim.x / PI_NXCD;
        c = x * span + j;
        c_end = min((x + 1) * span, n_chunks);
        if (c >= c_end) return;
    } else {
        long long chunk0, n_chunks;
        if (!pi_first_chunk<PI_BLOCK_EVAL>(s_end - s_begin, cpw, chunk0, n_chunks)) return;
        c = chunk0;
        c_end = min(chunk0 + cpw, n_chunks);
        c_step = 1;
    }
    long long sb = s_begin + c * PI_BLOCK_EVAL;
    unsigned int lane = min(tid, (unsigned int)(min(s_end - sb, (long long)PI_BLOCK_EVAL) - 1));
#if PT_AFFINE
    static_assert(PI_NA <= 64, "action table across the lanes of one wave");
    PiStateIn nxt = pi_load_state(V, policy, nullptr, sb, lane, false);
    float act_lanes = 0.0f;
#pragma unroll
    for (int j = 0; j < PI_NA; ++j) act_lanes = ((tid & 63u) == (unsigned int)j) ? tab[PI_TAB_ACT + j] : act_lanes;
#elif PT_REORDER
    // table loads first, then the policy stream: vmcnt counts in order, so the LDS stores and the workgroup's barrier
    // wait for L2-served table lines only while the first policy load (an HBM miss) is still in flight
    {
        constexpr int kPer = (PI_GRID.tab_len + PI_BLOCK_EVAL - 1) / PI_BLOCK_EVAL;
        float t[kPer];
#pragma unroll
        for (int j = 0; j < kPer; ++j) t[j] = tab[min(j * PI_BLOCK_EVAL + (int)tid, PI_GRID.tab_len - 1)];
        __builtin_amdgcn_sched_barrier(0);
        PiStateIn first = pi_load_state(V, policy, nullptr, sb, lane, false);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const int i = j * PI_BLOCK_EVAL + (int)tid;
            if (i < PI_GRID.tab_len) lds_tab[i] = t[j];
        }
        __syncthreads();
        nxt_init = first;
    }
    PiStateIn nxt = nxt_init;
#else
    PiStateIn nxt = pi_load_state(V, policy, nullptr, sb, lane, false);
    pi_stage_table<PI_BLOCK_EVAL>(tab, lds_tab);
    __syncthreads();
#endif
#if PT_PF_DIST > 0
    // touch the first chunk's policy lines of the workgroup that will take over this CU's slots half a lifetime or so
    // later (PT_PF_DIST groups ahead on this XCD), so that ITS prologue finds them in L2 instead of HBM
    int pf_word = 0;
    if (!PERSISTENT) {
        const long long n_chunks_all = (s_end - s_begin + PI_BLOCK_EVAL - 1) / PI_BLOCK_EVAL;
        const long long groups = (n_chunks_all + cpw - 1) / cpw;
        const long long span = (groups + PI_NXCD - 1) / PI_NXCD;
        const long long j = blockIdx.x / PI_NXCD, g2 = (blockIdx.x % PI_NXCD) * span + j + PT_PF_DIST;
        if (j + PT_PF_DIST < span && g2 < groups) {
            const long long sb2 = s_begin + g2 * cpw * PI_BLOCK_EVAL;
            const unsigned int lane2 = min(tid, (unsigned int)(min(s_end - sb2, (long long)PI_BLOCK_EVAL) - 1));
            pf_word = *pi_lane_ptr(policy + sb2, lane2);
        }
    }
#endif
    PT_STAMP(3);
    int it = 0;
    for (; c < c_end; c += c_step, ++it) {
        const PiStateIn cur = nxt;
        const long long sb_c = sb;
        const unsigned int lane_c = lane;
        const bool more = c + c_step < c_end;
        if (more) {
            sb += c_step * PI_BLOCK_EVAL;
            lane = min(tid, (unsigned int)(min(s_end - sb, (long long)PI_BLOCK_EVAL) - 1));
            nxt = pi_load_state(V, policy, nullptr, sb, lane, false);
        }
#if PT_TRACE
        const int k = it - it0;
        const bool rec = k == 0 || k == 1;
        if (rec) PT_STAMP2(0);                    // A: chunk begins, next chunk's inputs requested
#endif
        float x[PI_D], ns[PI_D], reward;
#if PT_AFFINE
        pt_state_coords((unsigned int)sb_c + lane_c, x);
        const float a = __int_as_float(__builtin_amdgcn_ds_bpermute(cur.action << 2, __float_as_int(act_lanes)));
#else
        pi_state_coords((unsigned int)sb_c + lane_c, lds_tab, x);
        const float a = lds_tab[PI_TAB_ACT + cur.action];
#endif
#if PT_TRACE
        if (rec) { if (more) PT_WAIT_BUT1(); else PT_WAIT_ALL(); PT_STAMP2(1); }   // B: this chunk's inputs are in registers
#endif
        bool done;
        pi_dynamics(x, a, ns, &reward, &done);
        float e = 0.0f;
        unsigned int base = 0u;
        float fr[PI_D];
#pragma unroll
        for (int d = 0; d < PI_D; ++d) fr[d] = 0.0f;
        if (!done) pi_locate(ns, base, fr);
#if PT_TRACE
        if (rec) PT_STAMP2(2);                    // C: dynamics + cell search issued
#endif
#if PT_LDSBOX
        // TIMING ONLY (results are wrong): what the gather would cost if the wave fetched the box of V its successors
        // fall into with COALESCED 16-byte loads into LDS (PT_LDSBOX dwords per state) and every lane then read its
        // 2^(D-1) corner pairs from there at lane-consecutive addresses — the shape the gather has once the lanes of a
        // wave share their successor cell along every dimension but the lane dimension (memory order, DESIGN.md s. 3).
        {
            constexpr int kSeg = PI_NPAIR, kLen = 72;                 // one segment per corner pair, 64 + slack floats
            __shared__ float lds_box[PI_BLOCK_EVAL / 64][kSeg * kLen];
            float* box = lds_box[tid >> 6];
            const unsigned int l = tid & 63u;
            const unsigned int base0 = __builtin_amdgcn_readfirstlane(base);
            constexpr int kLoads = (PT_LDSBOX * 64 + 255) / 256;      // wave-wide 16-byte loads
#pragma unroll
            for (int it = 0; it < kLoads; ++it) {
                unsigned int off = base0 + (unsigned int)((it * 64 + l) * 4);
                off = off < (unsigned int)(PI_GRID.n - 4) ? off : 0u;
                const float4 q = *reinterpret_cast<const float4*>(V + (off & ~3u));
                if ((it * 64 + l) * 4 + 3 < kSeg * kLen) *reinterpret_cast<float4*>(box + (it * 64 + l) * 4) = q;
            }
            __builtin_amdgcn_s_setprio(1);
            __builtin_amdgcn_wave_barrier();
            PiPair vp[PI_NPAIR];
            const unsigned int drift = (base - base0) & 7u;
#pragma unroll
            for (int m = 0; m < PI_NPAIR; ++m) {
                vp[m].x = box[m * kLen + l + drift];
                vp[m].y = box[m * kLen + l + drift + 1];
            }
            e = done ? 0.0f : pi_combine_corners(vp, fr);
            __builtin_amdgcn_s_setprio(0);
        }
        if (false) {
            PiPair vp[PI_NPAIR];
#else
        if (!done) {
            PiPair vp[PI_NPAIR];
            pi_request_corners(V, base, vp);
#endif
#if PT_TRACE
            if (rec) PT_STAMP2(3);                // D: the 2^(D-1) corner loads issued
#endif
#if PT_WPRIO == 0
            __builtin_amdgcn_s_setprio(1);
#elif PT_WPRIO == 1
            __builtin_amdgcn_s_setprio(3);
#elif PT_WPRIO == 2
            __builtin_amdgcn_s_setprio(2);
#endif
#if PT_TRACE
            if (rec) { PT_WAIT_ALL(); PT_STAMP2(4); }                              // E: corner values back
#endif
            e = pi_combine_corners(vp, fr);
#if PT_WPRIO == 0
            __builtin_amdgcn_s_setprio(0);
#elif PT_WPRIO != 3
            pt_setprio(base_prio);
#endif
        }
        const float nv = reward + gamma * e;
        if (tid == lane_c) pi_store_lane(Vn + sb_c, lane_c, nv);
#if PT_TRACE
        if (rec) PT_STAMP2(5);                    // F: weights + fmaf chain + store issued
#endif
    }
#if PT_PF_DIST > 0
    if (pf_word == 0x7fffffff) Vn[0] = 0.0f;          // keeps the touch alive; policy entries are action indices
#endif
#if PT_TRACE
    PT_STAMP(16);
    const unsigned int hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID
    const unsigned int xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
    if (((hw >> 8) & 0xfu) == 0u) {                                          // CU 0 of every shader array
        PT_PUT(hw, 0);
        PT_PUT(xcc, 1);
        PT_PUT(blockIdx.x, 17);
        PT_PUT(__builtin_amdgcn_readfirstlane(tid >> 6), 18);
        PT_PUT(it, 19);
        PT_PUT((unsigned int)__builtin_amdgcn_s_memrealtime(), 22);
        unsigned int slot0 = 0u;
        if ((tid & 63u) == 0u) slot0 = atomicAdd(trace_count, 1u);
        slot0 = __builtin_amdgcn_readfirstlane(slot0);
        if (slot0 < trace_cap) trace[(size_t)slot0 * PT_WORDS + (tid & 63u)] = (unsigned int)tr;
    }
#endif
}

extern "C" __global__ void __launch_bounds__(PI_BLOCK_EVAL) __attribute__((amdgpu_num_sgpr(80)))
pt_eval_kernel(const float* __restrict__ V, float* __restrict__ Vn, const int* __restrict__ policy,
               const float* __restrict__ tab, long long s_begin, long long s_end, float gamma, int cpw,
               unsigned int* __restrict__ trace, unsigned int* __restrict__ trace_count, unsigned int trace_cap, int it0) {
    pt_eval_body<0>(V, Vn, policy, tab, s_begin, s_end, gamma, cpw, trace, trace_count, trace_cap, it0);
}
extern "C" __global__ void __launch_bounds__(PI_BLOCK_EVAL) __attribute__((amdgpu_num_sgpr(80)))
pt_eval_persistent_kernel(const float* __restrict__ V, float* __restrict__ Vn, const int* __restrict__ policy,
               const float* __restrict__ tab, long long s_begin, long long s_end, float gamma, int cpw,
               unsigned int* __restrict__ trace, unsigned int* __restrict__ trace_count, unsigned int trace_cap, int it0) {
    pt_eval_body<1>(V, Vn, policy, tab, s_begin, s_end, gamma, cpw, trace, trace_count, trace_cap, it0);
}

// ---- experiment: one thread sweeps PT_G states that differ only in dimension PT_GDIM (tools/phase_timeline.py) ----
// Everything in step_dynamics that does not depend on x[PT_GDIM] (nor on the action) is loop-invariant and the compiler
// hoists it out of the loop over the PT_G states, as it does over the action loop of the improvement sweep.  A workgroup
// takes a "tile": PI_BLOCK_EVAL consecutive states inside one run of the dimensions behind PT_GDIM, times PT_G
// consecutive indices of PT_GDIM; lanes stay adjacent in memory, every policy load and V' store stays coalesced.
#ifndef PT_G
#define PT_G 4
#endif
#ifndef PT_GDIM
#define PT_GDIM 1
#endif
extern "C" __global__ void __launch_bounds__(PI_BLOCK_EVAL) __attribute__((amdgpu_num_sgpr(80)))
pt_eval_gloop_kernel(const float* __restrict__ V, float* __restrict__ Vn, const int* __restrict__ policy,
               const float* __restrict__ tab, long long s_begin, long long s_end, float gamma, int cpw,
               unsigned int* __restrict__ trace, unsigned int* __restrict__ trace_count, unsigned int trace_cap, int it0) {
    __shared__ float lds_tab[PI_GRID.tab_len];
    constexpr unsigned int kRun = (unsigned int)PI_GRID.stride[PT_GDIM];               // states behind PT_GDIM
    constexpr bool kFits = kRun % PI_BLOCK_EVAL == 0 && PI_GRID.g[PT_GDIM] % PT_G == 0;   // else: not launched
    constexpr unsigned int kPerRun = kFits ? kRun / PI_BLOCK_EVAL : 1, kBlocks = kFits ? PI_GRID.g[PT_GDIM] / PT_G : 1;
    if (!kFits) return;
    const unsigned int tiles = (unsigned int)(PI_GRID.n / ((long long)PI_BLOCK_EVAL * PT_G));
    const unsigned int span = gridDim.x / PI_NXCD;
    const unsigned int T = (blockIdx.x % PI_NXCD) * span + blockIdx.x / PI_NXCD;
    if (T >= tiles) return;
    const unsigned int outer = T / (kBlocks * kPerRun), gb = (T / kPerRun) % kBlocks, r = T % kPerRun;
    const unsigned int tid = threadIdx.x;
    // first state of the lane: index `outer` over the dimensions in front of PT_GDIM, gb * PT_G along it, r-th piece of the run
    const unsigned int s0 = (outer * (unsigned int)PI_GRID.g[PT_GDIM] + gb * PT_G) * kRun + r * PI_BLOCK_EVAL + tid;
    int act[PT_G];
#pragma unroll
    for (int k = 0; k < PT_G; ++k) act[k] = __builtin_nontemporal_load(policy + s0 + k * kRun);
    pi_stage_table<PI_BLOCK_EVAL>(tab, lds_tab);
    __syncthreads();
    float x[PI_D];
    pi_state_coords(s0, lds_tab, x);
    const int i_g = (int)(gb * PT_G);
#pragma unroll
    for (int k = 0; k < PT_G; ++k) {
        float ns[PI_D], reward;
        x[PT_GDIM] = lds_tab[PI_GRID.bins_off[PT_GDIM] + i_g + k];
        const float a = lds_tab[PI_TAB_ACT + act[k]];
        bool done;
        pi_dynamics(x, a, ns, &reward, &done);
        float e = 0.0f;
        if (!done) {
            unsigned int base;
            float fr[PI_D];
            pi_locate(ns, base, fr);
            e = pi_interpolate(V, base, fr);
        }
        Vn[s0 + k * kRun] = reward + gamma * e;
    }
}
'''

PHASES = ["A>B wait for inputs", "B>C dynamics + cell search", "C>D issue corner loads", "D>E wait for corner values",
          "E>F weights + fmaf chain + store"]


EXTRA_FLAGS: list = []


def build(src_text: str, trace: int, block: int, tmp: Path, tag: str, defines: str = ""):
    tag_f = re.sub(r"[^A-Za-z0-9_]+", "_", tag)
    src = tmp / f"pt_{tag_f}.hip"
    text, hits = re.subn(r"#define PI_BLOCK_EVAL \d+", f"#define PI_BLOCK_EVAL {block}", src_text, count=1)
    assert hits == 1
    src.write_text(f"#define PT_TRACE {trace}\n" + defines + text + PT_KERNELS)

    out = tmp / f"pt_{tag_f}.hsaco"
    res = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "--genco",
                          "-include", "hip/hip_runtime.h", "-Rpass-analysis=kernel-resource-usage", *EXTRA_FLAGS, str(src), "-o", str(out)],
                         capture_output=True, text=True)
    if res.returncode != 0:
        raise SystemExit(res.stderr[-4000:])
    usage, fn = {}, None
    for line in res.stderr.splitlines():
        if "Function Name:" in line:
            fn = line.split("Function Name:")[1].split()[0]
            usage[fn] = {}
        elif fn:
            for key in ("TotalSGPRs:", " VGPRs:", "ScratchSize", "Occupancy"):
                if key in line:
                    usage[fn][key.strip(" :")] = line.split(":")[-1].strip()
    return out, {k: v for k, v in usage.items() if k.startswith("pt_") or k == "pi_eval_sweep_kernel"}


def affine_fit(table: np.ndarray, max_exceptions: int = 2):
    """(a, b, exceptions) with float32(i * b + a) == table[i] (float64 multiply, float64 add, one conversion — what the
    kernel does) for every i outside `exceptions` = [(i, table[i])...], or None.  np.linspace tables are float64 affine
    sequences rounded to float32; their entries next to zero (the middle of a symmetric odd grid is a rounding residue
    like 4e-16) cannot be hit by any affine real function, hence the exceptions.  The admissible (a, b) form a convex
    polygon — one strip per entry: the reals that round to table[i] — whose a-width is concave in b: ternary search."""
    t = np.asarray(table, np.float32)
    g = len(t)
    t64 = t.astype(np.float64)
    up = np.nextafter(t, np.float32(np.inf)).astype(np.float64)
    dn = np.nextafter(t, np.float32(-np.inf)).astype(np.float64)
    hi, lo = t64 + 0.5 * (up - t64), t64 - 0.5 * (t64 - dn)
    i = np.arange(g, dtype=np.float64)
    use = np.abs(t64) > 1e-4 * np.abs(t64).max()
    if use.sum() < 2:
        return None
    iu, hu, lu = i[use], hi[use], lo[use]
    width = lambda b: (hu - iu * b).min() - (lu - iu * b).max()   # noqa: E731
    span = iu[-1] - iu[0]
    b_lo, b_hi = (lu[-1] - hu[0]) / span, (hu[-1] - lu[0]) / span
    for _ in range(200):
        m1, m2 = b_lo + (b_hi - b_lo) / 3.0, b_hi - (b_hi - b_lo) / 3.0
        if width(m1) < width(m2):
            b_lo = m1
        else:
            b_hi = m2
    b = 0.5 * (b_lo + b_hi)
    if width(b) <= 0.0:
        return None
    a = 0.5 * ((hu - iu * b).min() + (lu - iu * b).max())
    got = (i * b + a).astype(np.float32)              # float64 multiply, float64 add, one rounding to float32
    bad = np.flatnonzero(got.view(np.uint32) != t.view(np.uint32))
    if len(bad) > max_exceptions:
        return None
    return float(a), float(b), [(int(k), float(t[k])) for k in bad]


class Module:
    def __init__(self, path):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.mod = ctypes.c_void_p()
        assert self.hip.hipModuleLoad(ctypes.byref(self.mod), str(path).encode()) == 0

    def fn(self, name):
        f = ctypes.c_void_p()
        assert self.hip.hipModuleGetFunction(ctypes.byref(f), self.mod, name.encode()) == 0, name
        return f

    def launch(self, f, blocks, threads, vals):
        args = (ctypes.c_void_p * len(vals))(*[ctypes.cast(ctypes.byref(v), ctypes.c_void_p) for v in vals])
        rc = self.hip.hipModuleLaunchKernel(f, ctypes.c_uint(blocks), 1, 1, ctypes.c_uint(threads), 1, 1, 0,
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), args, None)
        assert rc == 0, rc


def timed(f, reps, warm=3):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    out = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        e1.synchronize()
        out.append(e0.elapsed_time(e1) / reps)
    return min(out)


def fold(rec: np.ndarray, lines: list, label: str):
    """rec: (m, 64) uint32 wave records -> text."""
    hw, xcc = rec[:, 0], rec[:, 1] & 0xF
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
    t = rec[:, 2:17].astype(np.int64)                   # slots 2..16
    its = rec[:, 19].astype(np.int64)

    def d(a, b):                                       # 32-bit wrapping difference of slots a -> b
        return ((rec[:, b].astype(np.int64) - rec[:, a].astype(np.int64)) & 0xFFFFFFFF)

    life = d(2, 16)
    lines.append(f"## {label}: {len(rec)} wave records from CU 0 of every shader array "
                 f"({len(set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist())))} CUs), chunks per wave {np.median(its):.0f}")
    lines.append(f"wave lifetime (entry -> end), shader cycles: median {np.median(life):.0f}  p10 {np.percentile(life, 10):.0f}  "
                 f"p90 {np.percentile(life, 90):.0f}")
    wv = rec[:, 18]
    if wv.max() > 0:
        trend = [f"{int(np.median(life[wv == w])):d}" for w in range(int(wv.max()) + 1) if (wv == w).any()]
        lines.append("median lifetime by wave index in the workgroup: " + " ".join(trend))
        endrel = {}
        for b in np.unique(rec[:, 17])[:4000]:
            idx = np.flatnonzero(rec[:, 17] == b)
            if len(idx) == int(wv.max()) + 1:
                e = ((rec[idx, 16].astype(np.int64) - rec[idx, 2].astype(np.int64).min()) & 0xFFFFFFFF)
                endrel.setdefault("spread", []).append(int(e.max() - e.min()))
                endrel.setdefault("wg_life", []).append(int(e.max()))
                endrel.setdefault("idle", []).append(float((e.max() - e).mean()))
        if endrel:
            lines.append(f"per workgroup: lifetime (first entry -> last end) median {np.median(endrel['wg_life']):.0f}, end spread median "
                         f"{np.median(endrel['spread']):.0f}, mean cycles a finished wave's slot waits for the workgroup's last wave "
                         f"{np.mean(endrel['idle']):.0f}")
    lines.append(f"entry -> kernel arguments loaded: median {np.median(d(2, 21)):.0f}  p90 {np.percentile(d(2, 21), 90):.0f}")
    lines.append(f"entry -> tables staged + barrier passed: median {np.median(d(2, 3)):.0f}  p90 {np.percentile(d(2, 3), 90):.0f}")
    tot = np.zeros(len(rec))
    for k in range(2):
        ok = rec[:, 4 + 6 * k] != 0
        if not ok.any():
            continue
        lines.append(f"traced chunk {k} ({ok.sum()} waves), shader cycles per phase: median / mean / p90")
        for p, name in enumerate(PHASES):
            v = d(4 + 6 * k + p, 5 + 6 * k + p)[ok]
            lines.append(f"    {name:36s} {np.median(v):8.0f} {v.mean():8.0f} {np.percentile(v, 90):8.0f}")
            tot[ok] += v
        v = d(4 + 6 * k, 9 + 6 * k)[ok]
        lines.append(f"    {'whole chunk A>F':36s} {np.median(v):8.0f} {v.mean():8.0f} {np.percentile(v, 90):8.0f}")
    # per SIMD: occupancy and phase census over time (absolute s_memtime is shared inside an XCD)
    groups = defaultdict(list)
    for i in range(len(rec)):
        groups[(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]), int(simd[i]))].append(i)
    occ_hist = np.zeros(17)
    census = np.zeros(7)                                # prologue, 5 phases, between/after
    samples = 0
    gaps = []
    for key, idx in groups.items():
        idx = np.array(idx)
        t0 = rec[idx, 2].astype(np.int64)
        base = t0.min()
        rel = lambda col: ((rec[idx, col].astype(np.int64) - base) & 0xFFFFFFFF)   # noqa: E731
        start, end = rel(2), rel(16)
        lo, hi = np.percentile(start, 15), np.percentile(end, 85)
        if hi <= lo:
            continue
        grid = np.linspace(lo, hi, 400)
        bounds = {c: rel(c) for c in range(2, 17)}
        for g in grid:
            res = (start <= g) & (g < end)
            occ_hist[min(int(res.sum()), 16)] += 1
            samples += 1
            for w in np.flatnonzero(res):
                ph = 6
                if g < bounds[3][w]:
                    ph = 0
                else:
                    for k in range(2):
                        if rec[idx[w], 4 + 6 * k] == 0:
                            continue
                        for p in range(5):
                            if bounds[4 + 6 * k + p][w] <= g < bounds[5 + 6 * k + p][w]:
                                ph = 1 + p
                census[ph] += 1
        # slot turnover: time from a wave's end to the next wave start on this SIMD (sorted pairing)
        s_sorted, e_sorted = np.sort(start), np.sort(end)
        n_res = int(np.median([((start <= g) & (g < end)).sum() for g in grid]))
        if n_res and len(s_sorted) > n_res:
            gaps.extend((s_sorted[n_res:] - e_sorted[:len(s_sorted) - n_res]).tolist())
    if samples:
        lines.append(f"per SIMD ({len(groups)} SIMDs, {samples} samples over the middle of the launch): resident waves "
                     f"mean {np.dot(occ_hist, np.arange(17)) / samples:.2f}; share of time with k waves resident: "
                     + " ".join(f"{k}:{occ_hist[k] / samples:.2f}" for k in range(17) if occ_hist[k] / samples >= 0.005))
        names = ["prologue (launch, table, barrier)"] + PHASES + ["untraced chunk / epilogue"]
        lines.append("mean waves per SIMD in each phase: " + "; ".join(f"{n}: {census[i] / samples:.2f}" for i, n in enumerate(names)))
    if gaps:
        gaps = np.array(gaps)
        lines.append(f"slot turnover (k-th wave end -> (k + resident)-th wave start on the same SIMD), cycles: median {np.median(gaps):.0f} "
                     f"mean {gaps.mean():.0f} p90 {np.percentile(gaps, 90):.0f}")
    lines.append("")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="double_pendulum_swingup")
    ap.add_argument("--bins", type=int, default=80)
    ap.add_argument("--out", default="gpurun_out/phase_timeline.txt")
    ap.add_argument("--persistent", default="1024x64,512x128,256x256,1024x32,512x64",
                    help="(threads per workgroup) x (workgroups per XCD) pairs for the persistent schedule")
    ap.add_argument("--compile-only", action="store_true", help="build every variant without a GPU and print its registers")
    ap.add_argument("--blocks", default="", help="extra (threads, cpw) pairs for the copy, e.g. 512x2,512x4,1024x4")
    ap.add_argument("--state", choices=["bench", "zero"], default="bench")
    ap.add_argument("--variants", default="", help="prologue variants of the copy, e.g. R1P0,R0P32,R1P32 (R: table loads before "
                    "the policy load; P: touch the policy lines of the workgroup P groups ahead); run for every --blocks geometry")
    ap.add_argument("--trace-variants", default="", help="variants (of the product geometry) to trace as well")
    ap.add_argument("--gloop", default="", help="G-loop experiment: (threads)x(G)x(dim) triples, e.g. 256x4x1,256x8x1,640x4x1")
    ap.add_argument("--timing-only", action="store_true",
                    help="allow grids with terminal states (the copies treat every state as live) and variants whose results "
                         "are wrong on purpose (L<n>: corner values from an LDS box filled with n coalesced dwords per state)")
    ap.add_argument("--extra-flags", default="", help="extra hipcc flags for the diagnostic kernels, space separated")
    args = ap.parse_args()

    EXTRA_FLAGS.extend(args.extra_flags.split())
    cls = envs.ENVS[args.env]
    if args.compile_only:
        from dynamicprogramming_amd import _native
        tables = [np.asarray(b, np.float32) for b in cls.bins_space(args.bins).values()]
        eng = _native.Engine(cls._D, [len(b) for b in tables], [b.min() for b in tables], [b.max() for b in tables], tables,
                             cls.ACTIONS, device=-1)
        text = eng.kernel_source(envs.dynamics_source(args.env))
        tmp = Path(tempfile.mkdtemp(prefix="pt_"))
        for blk in sorted({eng.info(11)} | {int(p.split("x")[0]) for p in args.persistent.split(",") if p}):
            for tr_on in (0, 1):
                print(blk, "trace" if tr_on else "plain", json.dumps(build(text, tr_on, blk, tmp, f"c{blk}_{tr_on}")[1]))
        fits = [affine_fit(b) for b in tables]
        print("affine fits", fits)
        for triple in [t for t in args.gloop.split(",") if t]:
            b, g, dim = (int(v) for v in triple.split("x"))
            print("gloop", triple, json.dumps(build(text, 0, b, tmp, f"g{b}_{g}_{dim}", f"#define PT_G {g}\n#define PT_GDIM {dim}\n")[1].get("pt_eval_gloop_kernel")))
        for v in [v for v in args.variants.split(",") if v]:
            m = re.fullmatch(r"(A?)R(\d)P(\d+)(?:W(\d))?(?:L(\d+))?", v)
            d = (f"#define PT_REORDER {m.group(2)}\n#define PT_PF_DIST {m.group(3)}\n#define PT_WPRIO {m.group(4) or 0}\n"
                 f"#define PT_LDSBOX {m.group(5) or 0}\n")
            if m.group(1):
                d += ("#define PT_AFFINE 1\n#define PT_AFF_A_INIT {" + ",".join(float(f[0]).hex() for f in fits) + "}\n"
                      "#define PT_AFF_B_INIT {" + ",".join(float(f[1]).hex() for f in fits) + "}\n")
            for tr_on in (0, 1):
                print(v, tr_on, json.dumps(build(text, tr_on, eng.info(11), tmp, f"v{v}{tr_on}", d)[1]["pt_eval_kernel"]))
        return
    solver = envs.make(args.env, args.bins, device="cuda:0")
    eng = solver._backend.engine
    n, nA = solver.n_states, solver.n_actions
    gamma = float(np.float32(solver.config.gamma))
    if not args.timing_only:
        assert solver.d_terminal_mask is None or not bool(solver.d_terminal_mask.any()), "traced copy: grids without terminal states"
    gen = torch.Generator(device="cpu").manual_seed(0)
    V = solver.d_value_function
    V[:n].copy_(torch.randn(n, generator=gen, dtype=torch.float32))
    solver.d_policy[:n].copy_(torch.randint(0, nA, (n,), generator=gen, dtype=torch.int32))
    solver.d_new_value_function.copy_(V)
    if args.state == "bench":
        for _ in range(2):
            solver._evaluation_sweeps(10, gamma)
            solver._improvement_sweep(gamma)
    V = solver.d_value_function
    pol = solver.d_policy
    block, cpw = eng.info(11), eng.info(3)
    src_text = eng.kernel_source(envs.dynamics_source(args.env))
    user_tables = [np.asarray(b, np.float32) for b in cls.bins_space(args.bins).values()]
    tab = torch.from_numpy(np.concatenate([np.asarray(solver.action_space, np.float32)] +
                                          [user_tables[d] for d in eng.order])).cuda()      # the engine's memory order
    Vref = torch.empty_like(V)
    Vout = torch.empty_like(V)
    stream = torch.cuda.current_stream().cuda_stream

    def product():
        eng.eval_sweep(V.data_ptr(), Vref.data_ptr(), pol.data_ptr(), 0, 0, n, gamma, 0, stream)

    lines = [f"# tools/phase_timeline.py  env {args.env} bins {args.bins} states {n}  product geometry {block} x {cpw}  state {args.state}",
             f"# device {torch.cuda.get_device_name(0)}  extra flags {EXTRA_FLAGS}  HIP_FORCE_DEV_KERNARG={os.environ.get('HIP_FORCE_DEV_KERNARG')}"]
    ms_product = timed(product, 20)
    lines.append(f"product kernel (C ABI)              {ms_product:.4f} ms")
    tmp = Path(tempfile.mkdtemp(prefix="pt_"))
    cap = 1 << 16
    trace = torch.zeros(cap * 64, dtype=torch.int32, device="cuda:0")
    count = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    summary = {"product_ms": ms_product}

    def run_variant(tag, trace_on, blk, cpw_v, persistent_w, it0, defines="", gloop=0):
        hsaco, usage = build(src_text, trace_on, blk, tmp, tag, defines)
        mod = Module(hsaco)
        name = "pt_eval_gloop_kernel" if gloop else "pt_eval_persistent_kernel" if persistent_w else "pt_eval_kernel"
        f = mod.fn(name)
        if gloop:
            tiles = n // (blk * gloop)
            blocks = max(8, 8 * ((tiles + 7) // 8))
        elif persistent_w:
            blocks = 8 * persistent_w
        else:
            chunks = (n + blk - 1) // blk
            groups = (chunks + cpw_v - 1) // cpw_v
            blocks = max(8, 8 * ((groups + 7) // 8))
        vals = [ctypes.c_void_p(V.data_ptr()), ctypes.c_void_p(Vout.data_ptr()), ctypes.c_void_p(pol.data_ptr()),
                ctypes.c_void_p(tab.data_ptr()), ctypes.c_longlong(0), ctypes.c_longlong(n), ctypes.c_float(gamma),
                ctypes.c_int(cpw_v), ctypes.c_void_p(trace.data_ptr()), ctypes.c_void_p(count.data_ptr()),
                ctypes.c_uint(cap), ctypes.c_int(it0)]
        run = lambda: mod.launch(f, blocks, blk, vals)   # noqa: E731
        Vout.zero_()
        ms = timed(run, 20)
        same = bool(torch.equal(Vout[:n], Vref[:n]))
        u = usage.get(name, {})
        lines.append(f"{tag:36s}{ms:.4f} ms   V' == product: {same}   {blocks} workgroups x {blk}   "
                     f"VGPRs {u.get('VGPRs')} SGPRs {u.get('TotalSGPRs')} scratch {u.get('ScratchSize [bytes/lane]', u.get('ScratchSize'))} occupancy {u.get('Occupancy [waves/SIMD]', u.get('Occupancy'))}")
        summary[tag] = {"ms": ms, "identical": same, "usage": u}
        print(lines[-1], flush=True)
        if trace_on:
            count.zero_()
            trace.zero_()
            run()
            torch.cuda.synchronize()
            m = min(int(count.item()), cap)
            rec = trace[: m * 64].cpu().numpy().view(np.uint32).reshape(m, 64)
            np.save(str(Path(args.out).with_suffix("")) + "_" + re.sub(r"[^A-Za-z0-9_]+", "_", tag) + ".npy", rec[:, :24])
            return rec
        return None

    product()
    torch.cuda.synchronize()
    run_variant(f"copy {block}x{cpw}", 0, block, cpw, 0, 0)
    rec = run_variant(f"traced {block}x{cpw}", 1, block, cpw, 0, 0)
    recs = [(f"traced {block}x{cpw} (product schedule)", rec)]
    geoms = [(block, cpw)]
    for pair in [p for p in args.blocks.split(",") if p]:
        b, c = (int(v) for v in pair.split("x"))
        run_variant(f"copy {b}x{c}", 0, b, c, 0, 0)
        geoms.append((b, c))

    fits = [affine_fit(user_tables[d]) for d in eng.order]

    def defs(v):
        m = re.fullmatch(r"(A?)R(\d)P(\d+)(?:W(\d))?(?:L(\d+))?", v)
        text = (f"#define PT_REORDER {m.group(2)}\n#define PT_PF_DIST {m.group(3)}\n#define PT_WPRIO {m.group(4) or 0}\n"
                f"#define PT_LDSBOX {m.group(5) or 0}\n")
        if m.group(1):
            assert all(f is not None and not f[2] for f in fits), "a bin table of this env is not an exact affine float64 sequence"
            text += ("#define PT_AFFINE 1\n#define PT_AFF_A_INIT {" + ",".join(float(f[0]).hex() for f in fits) + "}\n"
                     "#define PT_AFF_B_INIT {" + ",".join(float(f[1]).hex() for f in fits) + "}\n")
        return text
    for v in [v for v in args.variants.split(",") if v]:
        for b, c in geoms:
            run_variant(f"{v} {b}x{c}", 0, b, c, 0, 0, defs(v))
    for triple in [t for t in args.gloop.split(",") if t]:
        b, g, dim = (int(v) for v in triple.split("x"))
        run_variant(f"gloop {b} threads x G={g} along dim {dim}", 0, b, 1, 0, 0, f"#define PT_G {g}\n#define PT_GDIM {dim}\n", gloop=g)
    for v in [v for v in args.trace_variants.split(",") if v]:
        recs.append((f"{v} {block}x{cpw} traced", run_variant(f"{v} {block}x{cpw} traced", 1, block, cpw, 0, 0, defs(v))))
    first = True
    for pair in [p for p in args.persistent.split(",") if p]:
        b, w = (int(v) for v in pair.split("x"))
        run_variant(f"persistent {b} x W={w}", 0, b, 1, w, 0)
        if first:
            recs.append((f"persistent {b} x W={w}, traced iterations 20 and 21",
                         run_variant(f"persistent {b} x W={w} traced", 1, b, 1, w, 20)))
            first = False
    lines.append("")
    for label, r in recs:
        if r is not None and len(r):
            fold(r, lines, label)
    out = Path(args.out)
    out.parent.mkdir(parents=True, exist_ok=True)
    out.write_text("\n".join(lines) + "\n")
    print("\n".join(lines))
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
