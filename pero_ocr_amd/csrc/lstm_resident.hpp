// lstm_resident.hpp — a whole BiLSTM layer's recurrence in ONE launch: the hidden state is handed from step to step inside the
// chip, through the L2 of the XCD the cooperating workgroups sit on.
//
// lstm.hpp runs one launch per time step (the kernel boundary is the step barrier): a step costs ~7.8 us of which 1.1 us are
// MFMAs - the rest is launch gap and L2 round trips, and every step re-reads the unit group's 64 KB of W_hh.  That chain
// (2 T L steps) is what bounds pages of long lines (BASELINE config 5: 2 x 975 steps per launch = 15 ms against 5 ms of
// convolutions) and what the convolutions of the next launch pay for on config 2.
//
// Here a CLUSTER of H/16 workgroups owns one (16-line slice, direction) pair for the whole layer.  Member `ug` keeps its
// W_hh fragments (the four gates of 16 hidden units, all of K) in REGISTERS for all steps and the cell state c of its
// (line, unit) pairs in registers too; per step it
//   1. waits until the cluster's counter says every member has published step s-1,
//   2. reads the slice's h_{s-1} (16 lines x H) with L1-bypassing loads,
//   3. runs the same split-K MFMA GEMM + LDS reduction + gate arithmetic as lstm_step_kernel (bit-identical results),
//   4. stores its 16 x 16 piece of h_s, waits for the stores to be acknowledged, bumps the counter (and stores the layer output y).
// The hand-off is cheap only because all members of a cluster run on ONE XCD: plain stores are written through to that XCD's
// L2, `s_waitcnt vmcnt(0)` waits for L2's acknowledgement, the counter is an atomic EXECUTED IN THAT L2 (workgroup scope),
// polls and h reads bypass the L1 (sc1 / nt: L2-served).  Measured (tools/xcd_cluster_probe.hip, every word of every hand-off
// checked, idle and under a memory-bound background kernel): 1.1 us per step for up to 256 workgroups, 1.9 us for 512, zero
// stale words - against 3.4-24 us for the agent-scope release / acquire protocol that cross-XCD hand-offs need (round 2's
// persistent variants paid that and lost).  Measured and dropped: the data as its own flag (8-byte {value, step tag} pairs,
// no counter / acknowledgement wait / barrier) - every waiting wave then re-reads its 8 KB of state until the tags match, and
// that polling traffic costs more than the three waits it removes (11.7 k against 8.5 k cycles per step).
//
// HIP promises nothing about workgroup -> XCD placement; observed: block b runs on XCD b % 8.  The grid is laid out for that
// (cluster c = blocks {(c / 8 * UG + m) * 8 + c % 8}), but correctness does not depend on it: at start every member publishes the
// XCC_ID it really runs on (agent scope, once per launch); a cluster whose members do NOT share an XCD runs the same loop with
// the agent-scope protocol (release fence + agent atomic / acquire fence).  Every spin is bounded; a timeout sets an error word
// the host checks at collect time.  Deadlock freedom: a cluster only waits for its own members, workgroups are dispatched in
// block order, and a cluster's blocks are (UG * 8)-aligned runs of the grid - a resident cluster is complete or is the
// launch's last, partly dispatched group, which only waits for earlier (complete) clusters to retire.
#pragma once
#include <hip/hip_runtime.h>
#include "conv_igemm.hpp"
#include "lstm.hpp"

namespace pocr {

struct LstmResidentArgs {
    const float *xproj;      // [rows][8H]  (dir, gate, unit), as LstmStepArgs
    const float *whh_frag;   // [2][H/16][H/16][4][64][4], as LstmStepArgs
    const void *whh2;        // f16x2 fragments (lstm.hpp: lstm_gemm_f16x2) or NULL = fp32 MFMA
    float *hbuf;             // [clusters][2][16][H]: the slice's hidden state, ping-pong
    float *y;                // [rows][2H]
    unsigned *sync;          // [clusters][32], ZEROED before the launch: word 0 step counter; slice 0 of a group also: word 1 arrivals of the placement check, words 8..8+UG-1 the members' XCC ids + 1
    unsigned *err;           // [2]: [0] a spin timed out, [1] clusters that had to take the agent-scope protocol (diagnostic)
    const int32_t *line_T, *row_off, *slice_T;   // ragged batches, as LstmStepArgs (NULL: uniform T)
    int32_t n, npad, T;
    int32_t spin_limit;      // polls before a wait gives up
    int32_t force_agent;     // test hook (POCR_LSTM_FORCE_AGENT=1): take the cross-XCD protocol even when the cluster shares an XCD
    unsigned *xcc_census;    // test hook: [grid] XCC_ID + 1 of every block, or NULL
    int32_t y_p2;            // layer output in the P2 layout (as LstmStepArgs)
};

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
constexpr int kXccIdGetreg = (3 << 11) | (0 << 6) | 20;      // s_getreg_b32 HW_REG_XCC_ID, bits [3:0]

// H = 64 * KPW (64, 128, 256); UG = H / 16 members per cluster.
// SL: slices per workgroup.  A workgroup serves SL consecutive 16-line slices of one direction round-robin (slice 0 step s,
// slice 1 step s, ..., slice 0 step s + 1): the W_hh registers are shared, and while one slice's hand-off is in flight the
// workgroup computes the next slice - the hand-off latency disappears from the chain, and the launch needs 1 / SL of the
// workgroups (c2: 256 instead of 512, one per CU, which leaves the convolutions of the next launch two workgroups per CU).
template <int KPW, int SL>
__global__ __launch_bounds__(256, 2) void lstm_resident_kernel(LstmResidentArgs a) {
    constexpr int H = 64 * KPW, KGT = H / 16, UG = KGT;
    // ONE shared object (a second one makes hipcc drain the LDS-DMA queue before every ds_read, gemm_f16x2.hpp):
    // [part: wave x gate x lane x reg partial sums | hpre: the NEXT slice-step's h_{s-1}, 16 rows of H floats + 4 pad, landed by LDS-DMA | xl | flags]
    constexpr int HP = H + 4;                   // row pitch of the landing zone in floats: +16 bytes, so the 16 rows of an MFMA A fragment hit 16 different bank groups
    constexpr int PART_F = 4 * 4 * 64 * 4, HPRE_F = (SL >= 2 && H == 256) ? 16 * HP : 0;
    constexpr int XL_F = SL * 1024;             // xl: the gate pre-activations x of each slice's next step ([slice][wave][lane][4]: every wave lands and reads its own 1 KB)
    __shared__ float smem[PART_F + HPRE_F + XL_F + 4];
    float *part = smem, *hpre = smem + PART_F, *xl = smem + PART_F + HPRE_F;
    int &s_fast = *reinterpret_cast<int *>(smem + PART_F + HPRE_F + XL_F), &s_abort = *reinterpret_cast<int *>(smem + PART_F + HPRE_F + XL_F + 1);
    int &s_pref = *reinterpret_cast<int *>(smem + PART_F + HPRE_F + XL_F + 2);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int b = blockIdx.x, xcd = b & 7, k = b >> 3;
    const int group = (k / UG) * 8 + xcd, ug = k % UG;          // group = (SL slices, direction); its members share an XCD
    const int n_slices = a.npad / 16, n_sg = (n_slices + SL - 1) / SL;
    if (group >= 2 * n_sg) return;
    const int sg = group >> 1, dir = group & 1;
    // sync words of slice j of this group: cluster id = (sg * SL + j) * 2 + dir; the group's placement words live in slice 0's block
    unsigned *gsync = a.sync + (size_t)((sg * SL) * 2 + dir) * 32;
    int Tmax = 0;
#pragma unroll
    for (int j = 0; j < SL; ++j) {
        const int slice = sg * SL + j;
        if (slice < n_slices) Tmax = max(Tmax, a.slice_T ? a.slice_T[slice] : a.T);
    }
    if (Tmax <= 0) return;                                       // (uniform over the group)

    // ---- placement check (once): do all members of this group share an XCD?
    if (tid == 0) {
        const unsigned my = (unsigned)__builtin_amdgcn_s_getreg(kXccIdGetreg) + 1u;
        __hip_atomic_store(gsync + 8 + ug, my, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(gsync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0, ok = 1;
        while (__hip_atomic_load(gsync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)UG) {
            if (++spins > a.spin_limit) { ok = 0; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        int fast = a.force_agent ? 0 : 1;
        if (a.xcc_census) a.xcc_census[b] = my;
        for (int m = 0; m < UG; ++m) fast &= __hip_atomic_load(gsync + 8 + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == my;
        if (!ok) { __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        if (ok && !fast && ug == 0) __hip_atomic_fetch_add(a.err + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_fast = fast; s_abort = !ok;
    }
    __syncthreads();
    if (s_abort) return;
    const bool fast = s_fast != 0;

    // ---- per-thread roles: MFMA operand lane (li = line, kq) and epilogue thread (line i, unit u)
    const int u = tid & 15, i = tid >> 4;
    const int unit = ug * 16 + u;
    int Ti[SL], Ts[SL];
    int row0[SL];                                 // (first row of the thread's line: rows < 2^31)
    float cprev[SL];
#pragma unroll
    for (int j = 0; j < SL; ++j) {
        const int slice = sg * SL + j, line = slice * 16 + i;
        const bool have = slice < n_slices;
        Ts[j] = have ? (a.slice_T ? a.slice_T[slice] : a.T) : 0;
        Ti[j] = have && line < a.n ? (a.line_T ? a.line_T[line] : a.T) : 0;
        row0[j] = a.row_off ? a.row_off[min(line, a.n - 1)] : line * a.T;
        cprev[j] = 0.f;
    }
    // gate pre-activations of (slice j, step s): requested a full round ahead, so they are in registers when the gates need them
    // gate pre-activations x of (slice j, step s): copied into LDS by the load unit ONE ROUND AHEAD (global_load_lds: no result
    // registers - the compiler would put a drain of the whole load queue in front of their first use, and with it the latency
    // of the request just made: xproj is 302 MB, served by MALL / HBM).  One piece per wave: lane (line 4 wave + (lane >> 4),
    // gate (lane >> 2) & 3, quarter lane & 3) fetches four consecutive units, so thread (line i, unit u) later reads what its OWN
    // wave landed - no barrier, only that wave's counted vmcnt.  Finished / padding lines fetch a clamped row (never used):
    // every wave issues exactly one piece per call, which the counted waits rely on.
    auto load_x = [&](int j, int s) {
        const int tl = max(Ti[j], 1);
        const int t = min(max(dir == 0 ? s : Ti[j] - 1 - s, 0), tl - 1);
        // (gate slot = gate ^ (line & 1): the two lines a 32-lane group of read_x covers then sit 16 banks apart instead of on the same ones)
        const float *xp = a.xproj + (size_t)(row0[j] + t) * (8 * H) + (size_t)dir * 4 * H + (size_t)(((lane >> 2) & 3) ^ ((lane >> 4) & 1)) * H + ug * 16 + (lane & 3) * 4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)xp,
                                         (__attribute__((address_space(3))) void *)(xl + (j * 4 + wave) * 256), 16, 0, 0);
    };
    const int xb0 = wave * 256 + (lane >> 4) * 64 + (lane & 15) + 16 * ((lane >> 4) & 1);      // dword of gate slot 0 of thread (line, unit) in its wave's piece
    auto read_x = [&](int j, float (&x)[4]) {              // thread (i, u): gate g of unit u of line i
        // gate slot = gate ^ (line & 1): two per-lane bases that differ in one address bit, constant offsets otherwise
        x[0] = xl[j * 1024 + xb0]; x[1] = xl[j * 1024 + (xb0 ^ 16)]; x[2] = xl[j * 1024 + xb0 + 32]; x[3] = xl[j * 1024 + (xb0 ^ 16) + 32];
    };
    // W_hh fragments of this unit group, resident for the whole layer (the step kernel re-reads them every step)
    const bool f16 = a.whh2 != nullptr;
    LstmW2<KPW> w2;
    f32x4 bv[KPW][4];
    if (f16) {
        lstm_load_w_f16x2<KPW>(a.whh2, dir, ug, wave, lane, w2);
    } else {
        const f32x4 *wf = reinterpret_cast<const f32x4 *>(a.whh_frag) + ((size_t)(dir * KGT + ug) * KGT) * 4 * 64 + lane;
#pragma unroll
        for (int q = 0; q < KPW; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) bv[q][g] = wf[((size_t)(wave + 4 * q) * 4 + g) * 64];
    }
#pragma unroll
    for (int j = 0; j < SL; ++j) load_x(j, 0);
    const int src = (((i >> 2) * 16 + u) * 4) + ((i & 3) ^ ((u >> 3) * 2));      // D layout of the reduced gates (as lstm_step_kernel), halves swapped for u >= 8 (below)
    const bool hi8 = li >= 8;

    unsigned *pend = nullptr;                                  // sync word of the slice whose last state store is not yet published
    auto bump = [&](unsigned *sy) {                            // (everybody's stores are acknowledged and a barrier has been passed)
        if (tid == 0) {
            if (fast) {
                __hip_atomic_fetch_add(sy, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // executed in this XCD's L2
            } else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(sy, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    // A slice-step's stores (its piece of h_s, the layer output y) are ISSUED at the start of the next slice-step, behind that
    // step's wait: in front of a prefetched slice-step only the four x loads are then younger than the LDS-DMA pieces, and the
    // wait can be counted (vmcnt(4)) instead of a drain that would also sit out the x loads' HBM latency and the stores' acknowledgement.
    float d_hn = 0.f;
    float *d_hdst = nullptr;
    int d_row = 0;
    bool d_y = false;
    auto flush_stores = [&]() {
        if (d_hdst) { *d_hdst = d_hn; d_hdst = nullptr; }
        if (d_y) { lstm_store_y(a.y, (size_t)d_row, 2 * H, dir * H + unit, d_hn, a.y_p2 != 0); d_y = false; }
    };
    auto publish_pending = [&]() {
        flush_stores();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's state store is acknowledged by L2
        __syncthreads();                                       // ... and everybody's
        bump(pend);
        pend = nullptr;
    };
    // Look-ahead (SL >= 2, f16x2): while slice-step (j, s) computes, the state h_{s'-1} of the NEXT slice-step (j', s') of this
    // workgroup - published two or more slice-steps ago when several slices take turns - is copied into LDS by the load unit:
    // thread 0 requests that slice's counter at the start of the step (one L2 read, consumed behind the MFMAs), and if every
    // member has published, the four waves each issue 4 LDS-DMA pieces (one 1 KB row of h per piece) behind the step's barrier.
    // The next slice-step then starts with its operand in LDS: no counter poll, no L2 round trip for h (profiles/r03_lstm_resident.txt
    // section 8: wait 580 + h loads ~1500 of a slice-step's ~5000 cycles).  Not ready (a single slice left, the first step): the
    // blocking path below, as before.  Same values, same MFMAs: bit-identical.
    constexpr bool PREF = SL >= 2 && H == 256;       // (one 1 KB LDS-DMA piece = one row of h)
    bool have_pre = false;                                     // hpre holds h_{s-1} of the slice-step that starts now
    for (int s = 0; s < Tmax; ++s) {
#pragma unroll
        for (int j = 0; j < SL; ++j) {
            if (s >= Ts[j]) continue;                          // (uniform over the group: slice_T)
            // the slice-step after this one (iteration order), if any
            int ns = -1, nj = 0;
            if constexpr (PREF) {
#pragma unroll
                for (int q = SL - 1; q >= 0; --q) if (q > j && s < Ts[q]) { ns = s; nj = q; }
                if (ns < 0) {
#pragma unroll
                    for (int q = SL - 1; q >= 0; --q) if (s + 1 < Ts[q]) { ns = s + 1; nj = q; }
                }
            }
            const bool use_pre = have_pre;
            have_pre = false;
            const int cl = (sg * SL + j) * 2 + dir;
            unsigned *sync = a.sync + (size_t)cl * 32;
            float *hc = a.hbuf + (size_t)cl * 2 * 16 * H;     // [2][16][H]
            const bool live = s < Ti[j];
            const int t = dir == 0 ? s : Ti[j] - 1 - s;
            const int row = row0[j] + (live ? t : 0);
            // ---- a publication of THIS slice still pending (SL = 1, or the other slices of the group have finished): complete it now
            if (pend == sync) { publish_pending(); }
            if (s == 0) __syncthreads();                       // (`part`: the previous slice's gate reads)
            // ---- wait for h_{s-1} of the whole slice
            if (use_pre) {
                // this wave's four LDS-DMA pieces of h have landed (and every older x piece): everything but the ONE x piece requested behind them ...
                asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                __syncthreads();                                       // ... and everybody's
                flush_stores();
            } else {
                flush_stores();
            }
            if (!use_pre && s > 0) {
                if (tid == 0) {
                    const unsigned want = (unsigned)UG * (unsigned)s;
                    int spins = 0;
                    while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {   // sc1 load: served by L2
                        if (++spins > a.spin_limit) { __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_abort = 1; break; }
                        if (spins > 64) __builtin_amdgcn_s_sleep(1);
                    }
                    if (!fast) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                __syncthreads();
                if (s_abort) return;
            }
            if (!use_pre) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this slice-step's x piece (requested at the slice's previous turn) and the stores just issued
            float xcur[4];
            read_x(j, xcur);                                   // (before this slice's NEXT piece is requested, behind the barrier below, into the same 1 KB)
            // look-ahead poll of the next slice-step's counter: requested now, looked at behind the MFMAs
            unsigned look = 0u;
            const bool look_ok = PREF && f16 && ns > 0 && !(ns == s + 1 && nj == j);
            unsigned *nsync = a.sync + (size_t)((sg * SL + nj) * 2 + dir) * 32;
            if (look_ok && tid == 0) look = __hip_atomic_load(nsync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            f32x4 acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (s > 0 && f16 && use_pre) {
                constexpr int NB = 2 * KPW, QB = (NB + 3) / 4;
                f32x4 x0[QB], x1[QB];
#pragma unroll
                for (int q = 0; q < QB; ++q) {
                    const int blk = min(wave + 4 * q, NB - 1);
                    // landed kq-major (below): the 16-lane groups of a ds_read_b128 mix two kq values - with the row's natural order
                    // their 16-byte units met on the same banks (LDS bank-conflict share 0.38, profiles/r04_pmc_summary.json)
                    const f32x4 *pp = reinterpret_cast<const f32x4 *>(hpre + li * HP + (kq * 16 + blk * 2) * 4);
                    x0[q] = pp[0]; x1[q] = pp[1];
                }
                lstm_mfma_f16x2<KPW>(x0, x1, wave, w2, acc);
            } else if (s > 0 && f16) {
                lstm_gemm_f16x2<KPW, true>(hc + (size_t)(s & 1) * 16 * H + (size_t)li * H, wave, kq, w2, acc);
            } else if (s > 0) {
                const float *hrow = hc + (size_t)(s & 1) * 16 * H + (size_t)li * H;
                f32x4 av[KPW];
#pragma unroll
                for (int q = 0; q < KPW; ++q)                 // nt loads bypass the L1 (a plain load could return what this CU read two steps ago)
                    av[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(hrow + (wave + 4 * q) * 16 + kq * 4));
#pragma unroll
                for (int q = 0; q < KPW; ++q)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q][jj], bv[q][g][jj], acc[g], 0, 0, 0);
            }
            // (lines 8..15 store their halves swapped - a select on the VALUES, the address stays one register: the transposed reads
            // below then find lines u and u + 8 on different banks)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4 *>(&part[((wave * 4 + g) * 64 + lane) * 4]) =
                    hi8 ? (f32x4){acc[g][2], acc[g][3], acc[g][0], acc[g][1]} : acc[g];
            // The previous slice-step's state store is published HERE, one slice-step late: its L2 acknowledgement has had this
            // step's wait + GEMM to arrive, and the barrier that orders everybody's acknowledgement is the one the LDS reduction
            // needs anyway (before: acknowledgement + a barrier of its own behind every store, 400-2000 cycles of a ~5 k slice-step)
            if constexpr (PREF) {
                if (tid == 0) {
                    const bool ready = look_ok && look >= (unsigned)UG * (unsigned)ns;
                    if (ready && !fast) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    s_pref = ready ? 1 : 0;
                }
            }
            if (pend) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                   // (also: every wave has read its fragments of hpre - it may be overwritten)
            if (pend) { bump(pend); pend = nullptr; }
            if constexpr (PREF) {
                if (s_pref) {
                    // h_{ns-1} of slice nj: 16 rows of H floats = KPW * 4 pieces of 1 KB (KPW = 4: one row per piece), four per wave
                    const float *src = a.hbuf + (size_t)((sg * SL + nj) * 2 + dir) * 2 * 16 * H + (size_t)(ns & 1) * 16 * H;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int piece = wave * 4 + q;                    // = the row of h (H = 256: 1 KB)
                        // LDS unit l of the row (lane-linear landing) = the row's unit (blk = (l & 15) >> 1, kq = l >> 4, half = l & 1)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)piece * H + ((((lane & 15) >> 1) * 8 + (lane >> 4) * 2 + (lane & 1)) * 4)),
                                                         (__attribute__((address_space(3))) void *)(hpre + piece * HP), 16, 0, 2 /* nt: served by L2, like the direct loads */);
                    }
                    have_pre = true;
                }
            }
            asm volatile("" ::: "memory");                     // the x piece below stays BEHIND the h pieces (the counted wait assumes that order)
            load_x(j, s + 1);                                  // x of this slice's next step: due at this slice's next turn
            float gate[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float sum = part[(0 * 4 + g) * 256 + src];
                sum += part[(1 * 4 + g) * 256 + src];
                sum += part[(2 * 4 + g) * 256 + src];
                sum += part[(3 * 4 + g) * 256 + src];
                gate[g] = sum;
            }
            float hn = 0.f;                                    // finished / padding lines publish zeros (never used)
            if (live) {
                float cn;
                lstm_cell(gate, xcur, cprev[j], cn, hn);
                cprev[j] = cn;
            }
            d_hn = hn;
            if (s + 1 < Ts[j]) {
                d_hdst = hc + (size_t)((s + 1) & 1) * 16 * H + (size_t)i * H + unit;
                pend = sync;                                   // issued at the start of the next slice-step, published behind its GEMM (above)
            }
            d_y = live; d_row = row;
        }
    }
    flush_stores();
    if (pend) publish_pending();
}

}  // namespace pocr
