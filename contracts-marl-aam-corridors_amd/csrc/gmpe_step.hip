// gmpe_step.hip — host side of libgmpe.so: the C ABI of include/gmpe.h, state allocation, tile-shape selection, launches, the split
// big-E pipeline, hipGraph replay of per-step launch sequences, timing hooks, and the small streaming kernels around the fused one
// (adjacency expansion, edge compaction, rollout-buffer masks).
//
// The fused GraphMPE step / reset kernel lives in gmpe_kernel.h (template k_env<BLOCK, AP, SC, FL>) and is instantiated per
// scenario variant by gmpe_sc.hip. A workgroup (tile) owns G consecutive environments; everything the reference does for one
// env in one `MultiAgentGraphEnv.step` (multiagent/environment.py:1021-1063) + the worker's auto-reset
// (onpolicy/envs/env_wrappers.py:865-870) happens inside ONE launch:
//
//   load SoA state -> LDS | decode action + integrate (closed-form unicycle, or MPE soft-contact forces, pair-parallel) |
//   all-pairs distances | phase FSM + goal reach (parallel restatement of the sequential agent loop, SURVEY.md §8a
//   "ordered-visibility rule") | graph stores (adj [A,E,E], node_obs [A,E,F]: 16-byte coalesced) by waves 1.. while wave 0
//   does reward / done / info / write-back | optional reset (wave-cooperative rejection sampler).
//
// Launch shapes: gmpe_step = one launch of the step instantiation (FL 0 / 1); gmpe_rollout_steps / gmpe_step_many = ONE
// launch of the rollout instantiation (FL 2) for K steps, state carried in LDS / registers; handles on the split big-E path run
// k_env -> compact [N,E,E] scratch -> k_adj_expand -> [N,A,E,E], chunk-pipelined on side streams.
//
// HBM-bound by construction: per env-step the kernel reads ~1 KB of state and writes 4·A·(E² + F·E + D + 2) bytes of fp32
// observations; no MFMA (there is no contraction here). All geometry is fp64 with contraction OFF so thresholds see the
// reference's roundings; values are rounded to fp32 once, on store, like GraphReplayBuffer's float32 copy (graph_buffer.py:226).
#include "gmpe_kernel.h"

namespace gmpe {
// ---------------------------------------------------------------- learner-side edge set
// process_adj (onpolicy/algorithms/utils/gnn_new.py:329-358): mask = (adj < d) & (adj > 0) on fp32
// (inclusive=1 gives update_graph's `<=`, …_july.py:1660), edges in (batch,row,col) order, node ids
// offset by batch*E. Three launches: per-graph count (a wave per graph), chunked scan, ordered per-graph compaction.
__device__ __forceinline__ bool edge_pred(float v, float d, int inclusive) {
    return (inclusive ? v <= d : v < d) && v > 0.0f;
}
// one WAVE per graph (4 graphs per 256-thread block): no LDS, no barrier
__global__ __launch_bounds__(256) void k_edge_count(const float* __restrict__ adj, int B, int EE, float d, int inclusive, int32_t* __restrict__ counts) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    const float* g = adj + (size_t)b * EE;
    int c = 0;
    for (int q0 = 0; q0 < EE; q0 += 64) {
        const int q = q0 + lane;
        c += __popcll(__ballot(q < EE && edge_pred(g[q < EE ? q : 0], d, inclusive)));
    }
    if (lane == 0) counts[b] = c;
}
// exclusive scan of the per-graph counts: one 1024-thread block per chunk of 1024 graphs. The chunk's base is the sum
// of ALL preceding counts, re-reduced by the block itself (L2-resident, <= B reads: no atomics, no extra pass); inside
// the chunk a wave-shuffle scan + 16 wave totals.
__global__ __launch_bounds__(1024) void k_edge_scan(const int32_t* __restrict__ counts, int B, int32_t* __restrict__ offsets, int32_t* __restrict__ total) {
    __shared__ int wtot[16];
    __shared__ int wbase[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int first = blockIdx.x * 1024, b = first + t;
    int part = 0;
    for (int q = t; q < first; q += 1024) part += counts[q];
    for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
    const int c = b < B ? counts[b] : 0;
    int incl = c;
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    if (lane == 0) wbase[w] = part;
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    int pre = 0;
    for (int k = 0; k < 16; ++k) pre += wbase[k];
    for (int k = 0; k < w; ++k) pre += wtot[k];
    if (b < B) offsets[b] = pre + incl - c;
    if (blockIdx.x == gridDim.x - 1 && t == 1023) *total = pre + incl;
}
__global__ __launch_bounds__(256) void k_edge_write(const float* __restrict__ adj, int B, int E, float d, int inclusive,
                                                    const int32_t* __restrict__ offsets, int32_t* __restrict__ edge_index,
                                                    float* __restrict__ edge_attr, int cap) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    const int EE = E * E;
    const float* g = adj + (size_t)b * EE;
    int base = offsets[b];                                   // wave-uniform running position: (batch,row,col) order
    for (int q0 = 0; q0 < EE; q0 += 64) {
        const int q = q0 + lane;
        const float v = q < EE ? g[q] : 0.0f;
        const bool f = q < EE && edge_pred(v, d, inclusive);
        const unsigned long long bal = __ballot(f);
        const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
        if (f && pos < cap) {
            const int r = q / E, cc = q - r * E;
            edge_index[pos] = b * E + r;
            edge_index[cap + pos] = b * E + cc;
            edge_attr[pos] = v;
        }
        base += __popcll(bal);
    }
}

// The same edge set from the COMPACT adjacency [N,E,E] (one matrix per env; the A ego copies of the reference are aliases of it,
// …_july.py:1625, 1647-1648): each env's matrix is read ONCE per pass instead of A times and the A id-shifted copies of its edge list
// are emitted from registers — identical output to k_edge_count / k_edge_scan / k_edge_write on the materialised [N*A,E,E] tensor.
// Two launches: (1) per-env counts + per-block totals (a wave per env, 4 envs per block); (2) each block re-reduces the totals of the
// blocks before it (<= N/4 values, L2-resident; no atomics, no scan launch), adds its own earlier waves and its waves write. I64: int64 ids.
__global__ __launch_bounds__(256) void k_edgec_count(const float* __restrict__ adj, int N, int EE, float d, int inclusive,
                                                     int32_t* __restrict__ counts, int32_t* __restrict__ btot) {
    __shared__ int wc[4];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + w;                                     // one wave per env
    int c = 0;
    if (n < N) {
        const float* g = adj + (size_t)n * EE;
        for (int q0 = 0; q0 < EE; q0 += 512) {                            // 8 independent loads per lane in flight (the pass is latency-bound)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int q = q0 + u * 64 + lane; v[u] = q < EE ? g[q] : 0.0f; }
#pragma unroll
            for (int u = 0; u < 8; ++u) c += __popcll(__ballot(q0 + u * 64 + lane < EE && edge_pred(v[u], d, inclusive)));
        }
        if (lane == 0) counts[n] = c;
    }
    if (lane == 0) wc[w] = c;
    __syncthreads();
    if (threadIdx.x == 0) btot[blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}
template <bool I64>
__global__ __launch_bounds__(256) void k_edgec_write(const float* __restrict__ adj, int N, int A, int E, float d, int inclusive,
                                                     const int32_t* __restrict__ counts, const int32_t* __restrict__ btot,
                                                     void* __restrict__ edge_index, float* __restrict__ edge_attr, long long cap, int32_t* __restrict__ total) {
    __shared__ long long red[4];
    __shared__ int cnt[4];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, EE = E * E;
    const int n = blockIdx.x * 4 + w;
    long long part = 0;
    for (int q = threadIdx.x; q < (int)blockIdx.x; q += 256) part += btot[q];
    for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
    if (lane == 0) { red[w] = part; cnt[w] = n < N ? counts[n] : 0; }
    __syncthreads();
    long long base = red[0] + red[1] + red[2] + red[3];                  // edges of ONE copy of every env before this block
    for (int k = 0; k < w; ++k) base += cnt[k];
    const int c = cnt[w];
    if (n < N) {
        const float* g = adj + (size_t)n * EE;
        const long long b0 = base * A;                                    // (batch,row,col) order: the A copies of env n follow each other
        int run = 0;
        for (int q0 = 0; q0 < EE; q0 += 512) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int q = q0 + u * 64 + lane; v[u] = q < EE ? g[q] : 0.0f; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = q0 + u * 64 + lane;
                const bool f = q < EE && edge_pred(v[u], d, inclusive);
                const unsigned long long bal = __ballot(f);
                if (f) {
                    const int pe = run + __popcll(bal & ((1ull << lane) - 1ull));
                    const int r = q / E, cc = q - r * E;
                    for (int a = 0; a < A; ++a) {
                        const long long pos = b0 + (long long)a * c + pe;
                        if (pos < cap) {
                            const long long id0 = ((long long)n * A + a) * E;
                            if (I64) { static_cast<long long*>(edge_index)[pos] = id0 + r; static_cast<long long*>(edge_index)[cap + pos] = id0 + cc; }
                            else { static_cast<int32_t*>(edge_index)[pos] = (int32_t)(id0 + r); static_cast<int32_t*>(edge_index)[cap + pos] = (int32_t)(id0 + cc); }
                            edge_attr[pos] = v[u];
                        }
                    }
                }
                run += __popcll(bal);
            }
        }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) {              // wave 3 of the last block: everything before it + its own env
        const long long t = (base + c) * A;
        *total = t > 0x7fffffffLL ? 0x7fffffff : (int32_t)t;
    }
}

// GraphReplayBuffer.insert's mask rules (onpolicy/utils/graph_buffer.py:223-251, graph_mpe_runner.py:85-90, 395-405) from the step's
// dones: masks = 0 where done; active_masks = 0 where done unless every agent of the env is done. One thread per (env, agent).
__global__ __launch_bounds__(256) void k_masks(const uint8_t* __restrict__ done, int N, int A, float* __restrict__ masks, float* __restrict__ active) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= N * A) return;
    const int n = q / A;
    bool all = true;
    for (int a = 0; a < A; ++a) all = all && done[(size_t)n * A + a] != 0;
    const bool d = done[q] != 0;
    if (masks) masks[q] = d ? 0.0f : 1.0f;
    if (active) active[q] = (d && !all) ? 0.0f : 1.0f;
}

// Big-E path (c4 / c5: the A ego copies of the adjacency are 89-97 % of a step's bytes): the fused kernel writes the env's single
// E x E matrix into the handle's scratch and this kernel materialises [N,A,E,E] from it — the reference's per-agent adj arrays alias
// ONE matrix (…_july.py:1625, 1647-1648), so the expansion is an exact broadcast. A pure streaming kernel at full
// occupancy (no LDS, ~16 VGPRs): each lane loads one float4 of the matrix (read 1/A of the bytes written) and stores it to the A
// copies with the nontemporal hint; a wave's store covers 1 KB contiguous, consecutive workgroups consecutive 4 KB of the same copy.
// Store pattern (tools/expandbw.hip, profiles/r02_expandbw.log): OUTPUT-contiguous — blockIdx.x covers the A*nq float4 of one env's
// [A,E,E] block, blockIdx.y strides over the envs of the chunk — so the workgroups in flight write whole 0.6-4 MB blocks front to
// back like a fill (c4 784 us = 6.9 TB/s, c5 shard 1319 us = 6.5 TB/s; one lane per source float4 with A strided stores, the
// pattern the fused kernel uses, reaches 5.7 / 5.4). The source float4 is re-read per copy: L2 hits.
typedef float v4f_t2 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_adj_expand(const float* __restrict__ src, float* __restrict__ dst, int n_lo, int n_hi, uint32_t nq, int A, uint32_t m_nq) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x, per = (uint32_t)A * nq;
    if (j >= per) return;
    const uint32_t a = nq <= 1 ? j : __umulhi(j, m_nq), m = j - a * nq;              // exact: j * nq < 2^32 (checked by gmpe_create)
    const v4f_t2* s4 = reinterpret_cast<const v4f_t2*>(src);
    v4f_t2* d4 = reinterpret_cast<v4f_t2*>(dst);
    for (int n = n_lo + blockIdx.y; n < n_hi; n += gridDim.y)
        __builtin_nontemporal_store(s4[(size_t)n * nq + m], d4 + (size_t)n * per + j);
}
// scalar variant for E*E % 4 != 0 (odd E)
__global__ __launch_bounds__(256) void k_adj_expand1(const float* __restrict__ src, float* __restrict__ dst, int n_lo, int n_hi, uint32_t EE, int A) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x, per = (uint32_t)A * EE;
    if (j >= per) return;
    const uint32_t m = j % EE;
    for (int n = n_lo + blockIdx.y; n < n_hi; n += gridDim.y) dst[(size_t)n * per + j] = src[(size_t)n * EE + m];
}

}  // namespace gmpe

// =================================================================== host side / C ABI
using namespace gmpe;

struct gmpe_handle {
    gmpe_config c;
    int device;
    DevState s;
    int A, L, O, E, D, F;
    std::vector<void*> allocs;
    bool timing = false;
    std::vector<hipEvent_t> ev;      // pairs
    size_t ev_used = 0;
    double t_total_ms = 0;
    int64_t t_launches = 0;
    int block = 0;
    int G = 1;                       // envs per workgroup
    int ablate = 0;
    int nt = 0;
    int spec = 0;
    int ap = 0;                      // exact-size instantiation in use (0: run-time sizes)
    const double* ovr = nullptr;     // safety-filter hook slot (gmpe_set_control_override)
    const uint8_t* ovr_use = nullptr;
    int split = 0;                   // big-E path: k_env -> compact scratch -> k_adj_expand
    int roll = 1;                    // gmpe_step_many runs the persistent rollout kernel
    int G_roll = 1, block_roll = 256;   // tile shape of the rollout kernel (its own register budget -> its own residency)
    float* adj_scratch = nullptr;    // [N,E,E] (split path only)
    // split path: chunks of envs flow through two kernels on side streams — k_env(chunk c) on env_st[c & 1], k_adj_expand(chunk c)
    // on exp_st after it — so the HBM-bound expansion of one chunk overlaps the latency-bound fused kernel of the next
    int chunks = 1;
    hipStream_t env_st[2] = {nullptr, nullptr}, exp_st = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> ev_chunk;
    hipStream_t part_st[4] = {nullptr, nullptr, nullptr, nullptr};   // gmpe_step_many_envs: one side stream per env range (created on first use)
    hipEvent_t part_ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> ev_exp;      // expansion of chunk c finished (back-pressure on the k_env streams)
    int ahead = 0;                       // k_env may run at most this many chunks ahead of the expansion (0: unbounded)
    int xstep = 0;                       // gmpe_step_many on the split path: chain the steps' pipelines (no join between steps)
    int chunks_x = 1, ahead_x = 0;       // chunking / run-ahead bound of the chained pipeline
    int rowpairs = 0;                    // rollouts of the exact-size instantiations visit agent-row pairs only (distance_force_pass)
    int nfuse = 0;                       // doubles per env of the separate pair-force buffer (fused rollouts of exact-size navigation_graph tiles), 0: none
    int ramp = -1;                       // GMPE_RAMP, read once by gmpe_create (-1: the default rule of split_pipeline)
    int rollnt = -1;                     // GMPE_ROLLNT, read once by gmpe_create (-1: nontemporal rollout stores by slot volume)
    unsigned long long* stamps = nullptr;
    hipEvent_t region_ev[2] = {nullptr, nullptr};
    int32_t* edge_ws = nullptr;      // [cap_graphs] counts | [cap_graphs] offsets | [cap_graphs/1024+2] chunk sums
    size_t edge_ws_graphs = 0;
    int32_t* edgec_ws = nullptr;     // compact-adjacency edge kernels: [N] counts | [N/4] block totals
    size_t edgec_ws_n = 0;
    // hipGraphs of open-loop rollouts (gmpe_step_many_prepare): K kernel nodes replayed by one hipGraphLaunch
    struct StepGraph { const int32_t* actions; int32_t K, S; gmpe_outputs out; hipGraph_t graph; hipGraphExec_t exec; };
    std::vector<StepGraph> graphs;
    hipStream_t cap_stream = nullptr;   // capture happens on a private stream (the caller's may be the legacy default stream)
    bool capturing = false;
};

static thread_local std::string g_err;
static int fail(int code, const std::string& m) { g_err = m; return code; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(GMPE_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

// All persistent state lives in ONE slab (256-B aligned sub-arrays): a tile's ~30 state loads then touch a
// handful of pages instead of one page per field.
template <typename T>
static void slab_take(char* base, size_t& off, T** p, size_t n, int fill_byte, std::vector<std::pair<size_t, std::pair<size_t, int>>>& fills) {
    const size_t bytes = (n ? n : 1) * sizeof(T);
    *p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    fills.push_back({off, {bytes, fill_byte}});
    off = (off + bytes + 255) / 256 * 256;
}
// The separate pair-force buffer (F2, 2*A*A doubles per env) exists only in launches that can run the fused distance + next-step-force pass: the rollout
// instantiation k_env<256, 10, SC_NAV, 2> (FUSE_OK in gmpe_kernel.h). Every other launch — step kernels, run-time-size or single-wave rollouts — carves no
// F2 and is not charged its LDS (the c2 step tile at G = 6 sat 100 bytes under the 48 KiB cut-off because of it).
static int nfuse_of(const gmpe_handle* h, int block, int ap, int fl) { return (fl == 2 && block > 64 && ap == 10) ? h->nfuse : 0; }
// scenario variant of a validated config (gmpe_kernel.h)
static int sc_of(const gmpe_config& c) {
    switch (c.scenario) {
        case GMPE_SCENARIO_TUBE_JULY: return SC_JULY;
        case GMPE_SCENARIO_ROT_INV: return SC_ROT;
        case GMPE_SCENARIO_TWO_PHASE: return SC_TWO;
        case GMPE_SCENARIO_THREE_PHASE: return SC_THREE;
        default: return c.num_walls > 0 ? SC_NAV_WALLS : SC_NAV;
    }
}
static int sc_dispatch_occ(int sc, int block, int ap, size_t lds, int roll = 0, int g = 0) {
    switch (sc) {
        case SC_NAV: return max_tiles_per_cu<SC_NAV>(block, ap, lds, roll, g);
        case SC_NAV_WALLS: return max_tiles_per_cu<SC_NAV_WALLS>(block, ap, lds, roll, g);
        case SC_JULY: return max_tiles_per_cu<SC_JULY>(block, ap, lds, roll, g);
        case SC_ROT: return max_tiles_per_cu<SC_ROT>(block, ap, lds, roll, g);
        case SC_TWO: return max_tiles_per_cu<SC_TWO>(block, ap, lds, roll, g);
        default: return max_tiles_per_cu<SC_THREE>(block, ap, lds, roll, g);
    }
}
static hipError_t sc_dispatch_lds(int sc, int lds) {
    switch (sc) {
        case SC_NAV: return set_max_lds<SC_NAV>(lds);
        case SC_NAV_WALLS: return set_max_lds<SC_NAV_WALLS>(lds);
        case SC_JULY: return set_max_lds<SC_JULY>(lds);
        case SC_ROT: return set_max_lds<SC_ROT>(lds);
        case SC_TWO: return set_max_lds<SC_TWO>(lds);
        default: return set_max_lds<SC_THREE>(lds);
    }
}

// ---------------------------------------------------------------- learner side: node_obs rows from entity tables (gmpe_expand_node_obs)
// One thread per (env-step b, ego i, entity k). The arithmetic repeats stream_graph_fn's three row variants operation for operation (this translation unit
// is compiled with the same -ffp-contract=off), so the rows are bit-identical to what the engine writes: tests/test_gpu_gather.py compares them for every scenario x
// feature type. KIND 0: relative, F = 8 (…_july.py:1694-1771); 1: rot_inv family, F = 7 (rot_inv.py:1690-1766; two: goal = corridor exit, two_phase_graph.py:1405);
// 2: graph_feat_type 'global', F = 7 (…_july.py:1672-1691).
template <int KIND>
__global__ __launch_bounds__(256) void k_node_expand(const double* __restrict__ tab, float* __restrict__ out, long long total, int A, int L, int E, int W, int two,
                                                     long long n_in, long long n_out, long long off) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int k = (int)(t % E);
    const long long r1 = t / E;
    const int ei = (int)(r1 % A);
    const long long b = r1 / A;                                          // env-step index in the table: block * n_in + env
    const long long blk = b / n_in, n = b - blk * n_in;
    const double* T = tab + (size_t)b * W;
    const double* ex = T; const double* ey = T + E;
    const double* vox = T + 2 * E; const double* voy = vox + A; const double* vnx = voy + A; const double* vny = vnx + A;
    const bool kag = k < A;
    const int kk = kag ? k : 0;
    const bool post = k <= ei;                                           // agent k's re-drawn velocity is visible to ego ei iff k <= ei (ordered-visibility rule)
    const float typ = kag ? 0.0f : (k < A + L ? 1.0f : 2.0f);
    const size_t row = ((size_t)(blk * n_out + off + n) * A + ei) * E + k;
    if (KIND == 0) {
        const double kx = ex[k], ky = ey[k];
        const double kvox = kag ? vox[kk] : 0.0, kvoy = kag ? voy[kk] : 0.0, kvnx = kag ? vnx[kk] : 0.0, kvny = kag ? vny[kk] : 0.0;
        const double gx = kag ? ex[A + kk] : kx, gy = kag ? ey[A + kk] : ky;
        const double px = ex[ei], py = ey[ei], evx = vnx[ei], evy = vny[ei];
        float4* d = reinterpret_cast<float4*>(out + row * 8);
        d[0] = make_float4((float)((post ? kvnx : kvox) - evx), (float)((post ? kvny : kvoy) - evy), (float)(kx - px), (float)(ky - py));
        d[1] = make_float4((float)(gx - px), (float)(gy - py), kag ? 0.0f : 1.0f, typ);
    } else if (KIND == 1) {
        const double* cn = vny + A; const double* sn = cn + A;
        const float kx = (float)ex[k], ky = (float)ey[k];
        const float kvox = kag ? (float)vox[kk] : 0.0f, kvoy = kag ? (float)voy[kk] : 0.0f, kvnx = kag ? (float)vnx[kk] : 0.0f, kvny = kag ? (float)vny[kk] : 0.0f;
        const int wx = W - (E + 31) / 32 - 2;                              // two_phase_graph: exit x, y sit right before the mask words
        const float gxk = two ? (float)T[wx] : (kag ? (float)ex[A + kk] : 0.0f), gyk = two ? (float)T[wx + 1] : (kag ? (float)ey[A + kk] : 0.0f);
        const float apx = (float)ex[ei], apy = (float)ey[ei], avx = (float)vnx[ei], avy = (float)vny[ei];
        const double cs = cn[ei], s_ = sn[ei];
        const float rvx = (post ? kvnx : kvox) - avx, rvy = (post ? kvny : kvoy) - avy;
        const float rpx = kx - apx, rpy = ky - apy;
        double o0, o1, o2, o3, o4, o5;
        rot2(cs, s_, (double)rvx, (double)rvy, o0, o1);
        rot2(cs, s_, (double)rpx, (double)rpy, o2, o3);
        if (kag) rot2(cs, s_, (double)(gxk - apx), (double)(gyk - apy), o4, o5); else { o4 = o2; o5 = o3; }
        float* d = out + row * 7;
        d[0] = (float)o0; d[1] = (float)o1; d[2] = (float)o2; d[3] = (float)o3; d[4] = (float)o4; d[5] = (float)o5; d[6] = typ;
    } else {
        const float kx = (float)ex[k], ky = (float)ey[k];
        const float kvox = kag ? (float)vox[kk] : 0.0f, kvoy = kag ? (float)voy[kk] : 0.0f, kvnx = kag ? (float)vnx[kk] : 0.0f, kvny = kag ? (float)vny[kk] : 0.0f;
        const float gx = kag ? (float)ex[A + kk] : kx, gy = kag ? (float)ey[A + kk] : ky;
        float* d = out + row * 7;
        d[0] = post ? kvnx : kvox; d[1] = post ? kvny : kvoy; d[2] = kx; d[3] = ky; d[4] = gx; d[5] = gy; d[6] = typ;
    }
}

// The E x E adjacency of an env-step from its entity table: f32(sqrt(dx^2 + dy^2)) with delta = pos[min(r,c)] - pos[max(r,c)] (World.calculate_distances,
// core.py:600-624 — distance_trip's expression, so the bits are the engine's), zero diagonal, rows / columns of masked nodes zeroed (…_july.py:1627-1648).
// One thread per group of VEC consecutive entries of one matrix; `copies` = 1 writes [.., E, E], `copies` = A the materialised [.., A, E, E].
template <int VEC>
__global__ __launch_bounds__(256) void k_adj_from_table(const double* __restrict__ tab, float* __restrict__ out, long long total, int E, int W, int copies,
                                                        long long n_in, long long n_out, long long off) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int EE = E * E, per = EE / VEC;
    const long long b = t / per;
    const int q = (int)(t - b * per) * VEC;
    const long long blk = b / n_in, n = b - blk * n_in;
    const double* T = tab + (size_t)b * W;
    const double* ex = T; const double* ey = T + E;
    const double* mw = T + (W - (E + 31) / 32);
    float v[VEC];
#pragma unroll
    for (int u = 0; u < VEC; ++u) {
        const int r = (q + u) / E, c = (q + u) - r * E;
        const int lo = r < c ? r : c, hi = r < c ? c : r;
        const unsigned wr = (unsigned)mw[r >> 5], wc = (unsigned)mw[c >> 5];
        const bool masked = ((wr >> (r & 31)) | (wc >> (c & 31))) & 1u;
        const double dx = ex[lo] - ex[hi], dy = ey[lo] - ey[hi];
        v[u] = (r == c || masked) ? 0.0f : (float)sqrt(dx * dx + dy * dy);
    }
    float* dst = out + (size_t)(blk * n_out + off + n) * copies * EE + q;
    for (int a = 0; a < copies; ++a) {
        if (VEC == 4) *reinterpret_cast<float4*>(dst + (size_t)a * EE) = make_float4(v[0], v[1], v[2], v[3]);
        else dst[(size_t)a * EE] = v[0];
    }
}

extern "C" {

int gmpe_expand_adj(const gmpe_config* cfg, int device, const double* table_dev, int64_t num_blocks, int64_t envs_per_block,
                    float* adj_dev, int64_t out_envs_per_block, int64_t out_env_offset, int32_t copies, void* stream) {
    if (!cfg || !table_dev || !adj_dev) return fail(GMPE_ERR_INVALID_ARG, "gmpe_expand_adj: null argument");
    if (cfg->abi_version != GMPE_ABI_VERSION) return fail(GMPE_ERR_INVALID_ARG, "gmpe_config.abi_version mismatch");
    if (num_blocks < 0 || envs_per_block < 1 || out_env_offset < 0 || out_env_offset + envs_per_block > out_envs_per_block || copies < 1)
        return fail(GMPE_ERR_INVALID_ARG, "gmpe_expand_adj: the block's envs do not fit the output's env range");
    if (num_blocks == 0) return GMPE_OK;
    const int E = gmpe_num_entities(cfg), W = gmpe_entity_table_width(cfg);
    if (cfg->num_agents < 1 || cfg->num_agents > GMPE_MAX_AGENTS || E > GMPE_MAX_ENTITIES) return fail(GMPE_ERR_INVALID_ARG, "gmpe_expand_adj: config out of range");
    HIPCHK(hipSetDevice(device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int EE = E * E, vec = (EE & 3) == 0 ? 4 : 1;
    const long long total = (long long)num_blocks * envs_per_block * (EE / vec);
    if ((total + 255) / 256 > 0x7fffffffLL) return fail(GMPE_ERR_INVALID_ARG, "gmpe_expand_adj: too many entries for one launch (split the blocks)");
    const dim3 grid((unsigned)((total + 255) / 256));
    if (vec == 4) hipLaunchKernelGGL((k_adj_from_table<4>), grid, dim3(256), 0, st, table_dev, adj_dev, total, E, W, (int)copies, (long long)envs_per_block, (long long)out_envs_per_block, (long long)out_env_offset);
    else hipLaunchKernelGGL((k_adj_from_table<1>), grid, dim3(256), 0, st, table_dev, adj_dev, total, E, W, (int)copies, (long long)envs_per_block, (long long)out_envs_per_block, (long long)out_env_offset);
    HIPCHK(hipGetLastError());
    return GMPE_OK;
}

int gmpe_expand_node_obs(const gmpe_config* cfg, int device, const double* table_dev, int64_t num_blocks, int64_t envs_per_block,
                         float* node_obs_dev, int64_t out_envs_per_block, int64_t out_env_offset, void* stream) {
    if (!cfg || !table_dev || !node_obs_dev) return fail(GMPE_ERR_INVALID_ARG, "gmpe_expand_node_obs: null argument");
    if (cfg->abi_version != GMPE_ABI_VERSION) return fail(GMPE_ERR_INVALID_ARG, "gmpe_config.abi_version mismatch");
    if (num_blocks < 0 || envs_per_block < 1 || out_env_offset < 0 || out_env_offset + envs_per_block > out_envs_per_block)
        return fail(GMPE_ERR_INVALID_ARG, "gmpe_expand_node_obs: the block's envs do not fit the output's env range");
    if (num_blocks == 0) return GMPE_OK;
    const int A = cfg->num_agents, L = cfg->num_landmarks, E = gmpe_num_entities(cfg), W = gmpe_entity_table_width(cfg);
    if (A < 1 || A > GMPE_MAX_AGENTS || L < A || E > GMPE_MAX_ENTITIES) return fail(GMPE_ERR_INVALID_ARG, "gmpe_expand_node_obs: config out of range");
    HIPCHK(hipSetDevice(device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long total = (long long)num_blocks * envs_per_block * A * E;
    const dim3 grid((unsigned)((total + 255) / 256));
    if ((total + 255) / 256 > 0x7fffffffLL) return fail(GMPE_ERR_INVALID_ARG, "gmpe_expand_node_obs: too many rows for one launch (split the blocks)");
    const int two = cfg->scenario == GMPE_SCENARIO_TWO_PHASE;
    if (cfg->graph_feat_type == 1) hipLaunchKernelGGL((k_node_expand<2>), grid, dim3(256), 0, st, table_dev, node_obs_dev, total, A, L, E, W, two, (long long)envs_per_block, (long long)out_envs_per_block, (long long)out_env_offset);
    else if (cfg->scenario >= GMPE_SCENARIO_ROT_INV) hipLaunchKernelGGL((k_node_expand<1>), grid, dim3(256), 0, st, table_dev, node_obs_dev, total, A, L, E, W, two, (long long)envs_per_block, (long long)out_envs_per_block, (long long)out_env_offset);
    else hipLaunchKernelGGL((k_node_expand<0>), grid, dim3(256), 0, st, table_dev, node_obs_dev, total, A, L, E, W, two, (long long)envs_per_block, (long long)out_envs_per_block, (long long)out_env_offset);
    HIPCHK(hipGetLastError());
    return GMPE_OK;
}

int gmpe_abi_version(void) { return GMPE_ABI_VERSION; }
const char* gmpe_last_error(void) { return g_err.c_str(); }
int gmpe_obs_dim(const gmpe_config* c) {
    switch (c->scenario) {
        case GMPE_SCENARIO_TUBE_JULY: return 19;
        case GMPE_SCENARIO_TWO_PHASE: case GMPE_SCENARIO_THREE_PHASE: return 15;
        default: return 13;
    }
}
int gmpe_node_feats(const gmpe_config* c) { return (c->scenario >= GMPE_SCENARIO_ROT_INV || c->graph_feat_type == 1) ? 7 : GMPE_NODE_FEATS; }
int gmpe_num_entities(const gmpe_config* c) { return c->num_agents + c->num_landmarks + c->num_obstacles; }
int gmpe_entity_table_width(const gmpe_config* c) {
    return 2 * gmpe_num_entities(c) + 4 * c->num_agents + (c->scenario >= GMPE_SCENARIO_ROT_INV ? 2 * c->num_agents : 0) + (c->scenario == GMPE_SCENARIO_TWO_PHASE ? 2 : 0) +
           (gmpe_num_entities(c) + 31) / 32;
}

static int field_info(const gmpe_handle* h, int f, void** ptr, size_t* bytes) {
    const size_t N = h->c.num_envs, NA = N * h->A;
    const DevState& s = h->s;
    switch (f) {
        case GMPE_F_X: *ptr = s.x; *bytes = NA * 8; break;
        case GMPE_F_Y: *ptr = s.y; *bytes = NA * 8; break;
        case GMPE_F_S2: *ptr = s.s2; *bytes = NA * 8; break;
        case GMPE_F_S3: *ptr = s.s3; *bytes = NA * 8; break;
        case GMPE_F_P_DIST: *ptr = s.p_dist; *bytes = NA * 8; break;
        case GMPE_F_TIME: *ptr = s.time; *bytes = NA * 8; break;
        case GMPE_F_STATUS: *ptr = s.status; *bytes = NA; break;
        case GMPE_F_PREV_PHASE: *ptr = s.prev_phase; *bytes = NA * 4; break;
        case GMPE_F_PHASE_REACHED: *ptr = s.phase_reached; *bytes = NA * 4; break;
        case GMPE_F_COOLDOWN: *ptr = s.cooldown; *bytes = NA * 4; break;
        case GMPE_F_GOAL_TRACKER: *ptr = s.goal_tracker; *bytes = NA * 4; break;
        case GMPE_F_CURRENT_STEP: *ptr = s.current_step; *bytes = N * 4; break;
        case GMPE_F_RNG_CTR: *ptr = s.rng_ctr; *bytes = N * 8; break;
        case GMPE_F_TUBE: *ptr = s.tube; *bytes = N * GMPE_TUBE_STRIDE * 8; break;
        case GMPE_F_LANDMARKS: *ptr = s.landmarks; *bytes = N * h->L * 2 * 8; break;
        case GMPE_F_OBSTACLES: *ptr = s.obstacles; *bytes = N * h->O * 2 * 8; break;
        case GMPE_F_TIMES_REQUIRED: *ptr = s.times_required; *bytes = NA * 4; break;
        case GMPE_F_DISTS_TO_GOAL: *ptr = s.dists_to_goal; *bytes = NA * 4; break;
        case GMPE_F_DIST_LEFT: *ptr = s.dist_left; *bytes = NA * 4; break;
        case GMPE_F_GOAL_REACHED: *ptr = s.goal_reached; *bytes = NA * 4; break;
        case GMPE_F_N_AGENT_COLL: *ptr = s.n_agent_coll; *bytes = NA * 4; break;
        case GMPE_F_N_OBST_COLL: *ptr = s.n_obst_coll; *bytes = NA * 4; break;
        case GMPE_F_SPACING_VIOL: *ptr = s.spacing_viol; *bytes = NA * 4; break;
        case GMPE_F_STEPS_IN_CORR: *ptr = s.steps_in_corr; *bytes = NA * 4; break;
        case GMPE_F_CONFORMANCE: *ptr = s.conformance; *bytes = NA * 4; break;
        case GMPE_F_GOAL_MIN_TIME: *ptr = s.goal_min_time; *bytes = NA * 8; break;
        case GMPE_F_DELTA_SPACING: *ptr = s.delta_spacing; *bytes = N * 8; break;
        case GMPE_F_ERROR_FLAGS: *ptr = s.error_flags; *bytes = N * 4; break;
        case GMPE_F_PREV_PROJ: *ptr = s.prev_proj; *bytes = NA * 8; break;
        default: return fail(GMPE_ERR_INVALID_ARG, "unknown field id");
    }
    return GMPE_OK;
}

int gmpe_create(const gmpe_config* cfg, int device, gmpe_handle** out) {
    if (!cfg || !out) return fail(GMPE_ERR_INVALID_ARG, "null argument");
    if (cfg->abi_version != GMPE_ABI_VERSION) return fail(GMPE_ERR_INVALID_ARG, "gmpe_config.abi_version mismatch");
    const int E = gmpe_num_entities(cfg);
    if (cfg->num_envs < 1 || cfg->num_agents < 1 || cfg->num_agents > GMPE_MAX_AGENTS || cfg->num_landmarks < cfg->num_agents ||
        cfg->num_obstacles < 0 || cfg->num_walls < 0 || cfg->num_walls > GMPE_MAX_WALLS || E > GMPE_MAX_ENTITIES)
        return fail(GMPE_ERR_INVALID_ARG, "config out of range (agents<=64, entities<=160, walls<=8, landmarks>=agents)");
    if (cfg->scenario < GMPE_SCENARIO_NAVIGATION_GRAPH || cfg->scenario > GMPE_SCENARIO_THREE_PHASE)
        return fail(GMPE_ERR_UNSUPPORTED, "unknown scenario");
    if ((cfg->scenario != GMPE_SCENARIO_NAVIGATION_GRAPH) == (cfg->dynamics == GMPE_DYN_DOUBLE_INTEGRATOR))
        return fail(GMPE_ERR_UNSUPPORTED, "the tube scenarios are kinematic; navigation_graph is double_integrator");
    if (cfg->graph_feat_type < 0 || cfg->graph_feat_type > 1) return fail(GMPE_ERR_UNSUPPORTED, "graph_feat_type: 0 (relative) or 1 (global)");
    if (cfg->formation_type < GMPE_FORMATION_POINT || cfg->formation_type > GMPE_FORMATION_CIRCLE)
        return fail(GMPE_ERR_UNSUPPORTED, "formation_type: 0 (point), 1 (line) or 2 (circle)");
    if (cfg->contact_family < 0 || cfg->contact_family > 1 || (cfg->contact_family == 1 && (cfg->scenario != GMPE_SCENARIO_NAVIGATION_GRAPH || !(cfg->agent_mass > 0))))
        return fail(GMPE_ERR_UNSUPPORTED, "contact_family 1 (classic MPE) applies to navigation_graph and needs agent_mass > 0");
    if (cfg->dynamics == GMPE_DYN_DOUBLE_INTEGRATOR ? (cfg->n_actions != 5 && cfg->n_actions != 9) : cfg->n_actions != 25)
        return fail(GMPE_ERR_INVALID_ARG, "n_actions does not match the dynamics");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(GMPE_ERR_NO_DEVICE, "no HIP device visible: the engine has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(GMPE_ERR_INVALID_ARG, "bad device index");
    HIPCHK(hipSetDevice(device));
    gmpe_handle* h = new gmpe_handle();
    h->c = *cfg; h->device = device;
    h->A = cfg->num_agents; h->L = cfg->num_landmarks; h->O = cfg->num_obstacles; h->E = E; h->D = gmpe_obs_dim(cfg); h->F = gmpe_node_feats(cfg);
    const size_t N = cfg->num_envs, NA = N * h->A;
    DevState& s = h->s;
    std::vector<std::pair<size_t, std::pair<size_t, int>>> fills;
    char* slab = nullptr;
    for (int pass = 0; pass < 2; ++pass) {                  // pass 0 sizes the slab, pass 1 hands out the pointers
        size_t off = 0;
        fills.clear();
#define AL(p, n, fill) slab_take(slab, off, &(p), (n), (fill), fills);
        AL(s.x, NA, 0) AL(s.y, NA, 0) AL(s.s2, NA, 0) AL(s.s3, NA, 0) AL(s.p_dist, NA, 0) AL(s.time, NA, 0)
        AL(s.status, NA, 0) AL(s.prev_phase, NA, 0) AL(s.phase_reached, NA, 0) AL(s.cooldown, NA, 0)
        AL(s.goal_tracker, NA, 0xFF) AL(s.current_step, N, 0) AL(s.rng_ctr, N, 0)
        AL(s.tube, N * GMPE_TUBE_STRIDE, 0) AL(s.landmarks, N * h->L * 2, 0) AL(s.obstacles, N * h->O * 2, 0)
        AL(s.times_required, NA, 0xFF) AL(s.dists_to_goal, NA, 0xFF) AL(s.dist_left, NA, 0xFF) AL(s.goal_reached, NA, 0xFF)
        AL(s.n_agent_coll, NA, 0) AL(s.n_obst_coll, NA, 0) AL(s.spacing_viol, NA, 0) AL(s.steps_in_corr, NA, 0)
        AL(s.conformance, NA, 0) AL(s.goal_min_time, NA, 0) AL(s.delta_spacing, N, 0) AL(s.error_flags, N, 0) AL(s.prev_proj, NA, 0)
#undef AL
        if (pass == 0) {
            void* q = nullptr;
            hipError_t e = hipMalloc(&q, off);
            if (e != hipSuccess) { gmpe_destroy(h); return fail(GMPE_ERR_HIP, std::string("hipMalloc(state slab): ") + hipGetErrorString(e)); }
            h->allocs.push_back(q);
            slab = static_cast<char*>(q);
        }
    }
    for (auto& f : fills) {
        hipError_t e = hipMemset(slab + f.first, f.second.second, f.second.first);
        if (e != hipSuccess) { gmpe_destroy(h); return fail(GMPE_ERR_HIP, std::string("hipMemset: ") + hipGetErrorString(e)); }
    }
    s.tape = nullptr; s.tape_len = 0;
    // Tile shape. G envs per workgroup so that the sequential-semantics passes fill wave 0 (G*A <= 64)
    // while the per-tile LDS stays small enough for several workgroups per CU; BLOCK threads share the
    // distance pass and the stores. GMPE_G / GMPE_BLOCK override the heuristic (tuning, tests).
    const char* env_g = getenv("GMPE_G");
    const char* env_block = getenv("GMPE_BLOCK");
    // Heuristic (measured on MI355X, profiles/README.md "tile shapes"): every tile should be RESIDENT at once (a second round of
    // tiles repeats the whole latency chain), with about four tiles per CU — one wave-0 chain per SIMD — and as many waves per
    // tile as still fit: C2/C3 pick G = 4, BLOCK = 256 (1024 tiles = 4 per CU x 4 waves at <= 128 VGPRs). Residency is asked
    // from the runtime (registers + LDS of the instantiation that will run). When the batch cannot be resident at all
    // (C4/C5, N >> 4096) a tile packs as many envs as wave 0 holds (G*A <= 64) and the stores decide.
    int dev_cus = 256;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) dev_cus = prop.multiProcessorCount; }
    // Rollouts of the exact-size navigation_graph instantiation (A = L = 10, no obstacles, no walls) compute the next step's contact forces inside the distance
    // pass (distance_force_pass): the forces then need a buffer of their own, 2*A*A doubles per env (1.6 KB). GMPE_FUSE=0: the classic force pass.
    h->nfuse = (sc_of(h->c) == SC_NAV && h->A == h->L && h->O == 0 && h->A == 10 && !(getenv("GMPE_AP") && atoi(getenv("GMPE_AP")) == 0)) ? 2 * h->A * h->A : 0;
    if (getenv("GMPE_FUSE") && atoi(getenv("GMPE_FUSE")) == 0) h->nfuse = 0;
    h->rowpairs = (getenv("GMPE_FUSE") && atoi(getenv("GMPE_FUSE")) == 0) ? 0 : 1;
    int Gmax = 64 / h->A; if (Gmax < 1) Gmax = 1; if (Gmax > (int)N) Gmax = (int)N;
    while (Gmax > 1 && lds_bytes(Gmax, h->A, E, h->D, cfg->num_walls, h->nfuse) > 48 * 1024) --Gmax;   // one cap for both tile shapes (the rollout tile carries the F2 buffer)
    int ap_sel = (h->A == h->L && h->O == 0 && (h->A == 10 || h->A == 3)) ? h->A : 0;   // exact-size instantiations of the common cases (A = L, no obstacles)
    if (getenv("GMPE_AP") && atoi(getenv("GMPE_AP")) == 0) ap_sel = 0;                    // tuning: run-time sizes
    h->ap = ap_sel;
    int G = 0, block_sel = 0;
    if (!env_g && !env_block) {
        int G0 = (int)((N + 4 * (size_t)dev_cus - 1) / (4 * (size_t)dev_cus));
        if (G0 < 1) G0 = 1;
        const int blocks[3] = {256, 128, 64};
        for (int bi = 0; bi < 3 && !G; ++bi)
            for (int g = G0; g <= Gmax; ++g) {
                const size_t tiles = (N + g - 1) / g;
                const int per_cu = sc_dispatch_occ(sc_of(h->c), blocks[bi], ap_sel, lds_bytes(g, h->A, E, h->D, cfg->num_walls, 0), 0, g);
                if (per_cu > 0 && tiles <= (size_t)per_cu * dev_cus) { G = g; block_sel = blocks[bi]; break; }
            }
    }
    if (!G) {
        G = env_g ? atoi(env_g) : Gmax;
        if (G < 1) G = 1;
        if (G > Gmax) G = Gmax;
    }
    {   // exact magic division (fdiv) needs q*d < 2^32 for every (range, divisor) pair the kernel uses
        const uint64_t S = (uint64_t)h->L + h->O;
        uint64_t d = (uint64_t)h->A * E;
        if ((uint64_t)h->A * (E * E / 4) * (E * E / 4) >= (1ull << 32)) { gmpe_destroy(h); return fail(GMPE_ERR_UNSUPPORTED, "adjacency block too large for the index arithmetic"); }
        const uint64_t cand[] = {(uint64_t)E * E, S * S, (uint64_t)h->A * (h->A + h->O), (uint64_t)h->A * h->D, 2ull * E};
        for (uint64_t x : cand) if (x > d) d = x;
        if ((uint64_t)Gmax * d * d >= (1ull << 32)) { gmpe_destroy(h); return fail(GMPE_ERR_UNSUPPORTED, "tile too large for the index arithmetic"); }
    }
    h->G = G;
    {   // rollout kernel: same rule (every tile resident, from four tiles per CU upward) with ITS registers; BLOCK 256 unless the step kernel runs single-wave tiles
        const char* env_gr = getenv("GMPE_GROLL");
        h->block_roll = ((env_block ? atoi(env_block) : block_sel) == 64) ? 64 : 256;
        int Gr = 0;
        if (!env_g && !env_gr) {
            int G0 = (int)((N + 4 * (size_t)dev_cus - 1) / (4 * (size_t)dev_cus));
            if (G0 < 1) G0 = 1;
            for (int g = G0; g <= Gmax && !Gr; ++g) {
                const size_t tiles = (N + g - 1) / g;
                const int per_cu = sc_dispatch_occ(sc_of(h->c), h->block_roll, ap_sel, lds_bytes(g, h->A, E, h->D, cfg->num_walls, nfuse_of(h, h->block_roll, ap_sel, 2)), 1, g);
                if (per_cu > 0 && tiles <= (size_t)per_cu * dev_cus) Gr = g;
            }
        }
        if (!Gr) Gr = env_gr ? atoi(env_gr) : (env_g ? G : Gmax);
        if (Gr < 1) Gr = 1;
        if (Gr > Gmax) Gr = Gmax;
        h->G_roll = Gr;
    }
#if defined(GMPE_DIAG) || defined(GMPE_NTORDER_KNOB)
    h->ablate = getenv("GMPE_ABLATE") ? atoi(getenv("GMPE_ABLATE")) : 0;   // diagnostic build only: timing ablations (wrong results)
#else
    h->ablate = 0;
#endif
    // Graph outputs per launch vs the 256 MiB Infinity Cache: small launches (C2/C3: 92 MB) are absorbed by it and run
    // faster with ordinary stores (37.0 vs 40.4 us measured); big ones (C4 6 GB, C5 9 GB) stream past it and gain ~10 %
    // from nontemporal stores (C4 1417 -> 1285 us, C5 2005 -> 1852 us).
    {
        const double out_bytes = (double)N * h->A * ((double)E * E + 8.0 * E) * 4.0;
        h->nt = getenv("GMPE_NT") ? atoi(getenv("GMPE_NT")) : (out_bytes > 192.0 * 1024 * 1024 ? 1 : 0);
        // Split path (fused kernel -> compact matrix, k_adj_expand -> A copies, chunk-pipelined; see split_pipeline): pays where the
        // adjacency is >= ~95 % of the bytes (A >= 48): c5 shapes (64 agents) 2048 envs 1515 us split vs 1731 fused, 4096: 2994 vs 3455,
        // 8192: 6043 vs 6565, 16384: 11618 vs 12981 (with the run-ahead bound below); c4 (A = 32) 1260 vs 1132: stays fused
        // (profiles/r02_notes.md). Everything else runs the fused kernel.
        const size_t tiles_resident = (size_t)dev_cus * (size_t)[&] { const int q = sc_dispatch_occ(sc_of(h->c), block_sel ? block_sel : 256, ap_sel, lds_bytes(G, h->A, E, h->D, cfg->num_walls, 0), 0, G); return q > 0 ? q : 1; }();
        const size_t tiles_total = (N + G - 1) / G;
        h->split = getenv("GMPE_SPLIT") ? atoi(getenv("GMPE_SPLIT"))
                                        : (out_bytes > 192.0 * 1024 * 1024 && h->A >= 48 ? 1 : 0);
        if ((uint64_t)N * E * E >= (1ull << 32)) h->split = 0;            // k_adj_expand indexes the compact matrix with 32 bits
        h->roll = getenv("GMPE_ROLL") ? atoi(getenv("GMPE_ROLL")) : 1;
        h->rollnt = getenv("GMPE_ROLLNT") ? (atoi(getenv("GMPE_ROLLNT")) != 0) : -1;
        if ((uint64_t)h->A * E * E / 4 * (E * E / 4 + 1) >= (1ull << 32)) h->split = 0;   // k_adj_expand's exact magic division
        if (h->split) {
            void* q = nullptr;
            hipError_t e = hipMalloc(&q, (size_t)N * E * E * sizeof(float));
            if (e != hipSuccess) { gmpe_destroy(h); return fail(GMPE_ERR_HIP, std::string("hipMalloc(adjacency scratch): ") + hipGetErrorString(e)); }
            h->allocs.push_back(q);
            h->adj_scratch = static_cast<float*>(q);
            // Back-pressure: the fused kernel is the faster stage, and left alone it runs many chunks ahead; by the time k_adj_expand reads a
            // chunk's matrices they have left the 256 MiB Infinity Cache, and its A re-reads per matrix (one per ego copy, spread over the
            // XCDs' L2s) come from HBM: 220 us per 256-env chunk instead of 150 (rocprofv3 trace, profiles/r02_notes.md). With k_env(c)
            // waiting for expand(c - 2) the live part of the scratch stays cache-resident: c5 shapes 4096 envs 3640 -> 2994 us, 8192
            // 7744 -> 6043, 16384 16003 -> 11618. A scratch that fits the cache as a whole (the 2048-env shard: 134 MB) needs no bound.
            h->ahead = (double)N * E * E * sizeof(float) > 192.0 * 1024 * 1024 ? 2 : 0;
            if (getenv("GMPE_AHEAD")) h->ahead = atoi(getenv("GMPE_AHEAD"));
            if (h->ahead < 0) h->ahead = 0;
            // gmpe_step_many (open-loop steps): the steps' pipelines are chained — k_env(step s+1, chunk c) waits for expand(step s, chunk c), not for
            // the whole step s — which hides the pipeline's fill: c5 shard 1486-1524 -> 1383 us per step, 16384 envs 11.26 -> 10.94 ms. The chained
            // pipeline is long at any batch size, so it always uses two rounds of tiles per chunk and the run-ahead bound of 2 (shard: 8 chunks
            // unbounded 1416, 4 chunks D = 2 1383, 3 / 5 / 6 chunks 1409 / 1410 / 1434; D = 1 1982, D = 3 1407).
            h->xstep = getenv("GMPE_XSTEP") ? atoi(getenv("GMPE_XSTEP")) : 1;
            h->ramp = getenv("GMPE_RAMP") ? (atoi(getenv("GMPE_RAMP")) != 0) : -1;
            h->chunks_x = (int)((tiles_total + 2 * tiles_resident - 1) / (2 * tiles_resident));
            if (h->chunks_x < 2) h->chunks_x = 2;
            if (h->chunks_x > 128) h->chunks_x = 128;
            h->ahead_x = 2;
            // one chunk = one full round of resident tiles (c5 shard of 2048 envs: 8 chunks of 256 envs; 2 / 4 / 6 / 8 / 12 / 16 chunks: 1802 / 1602 /
            // 1560 / 1522 / 1548 / 1539 us), two rounds when the run-ahead is bounded (long pipelines: fewer launch gaps on the expansion
            // stream; 512- vs 256-env chunks: 4096 envs 2944-3029 vs 3104 us, 8192: 5729 vs 5828, 16384: 11223 vs 11521)
            const size_t per_chunk = tiles_resident * (h->ahead > 0 ? 2 : 1);
            h->chunks = (int)((tiles_total + per_chunk - 1) / per_chunk);
            if (h->chunks < 4) h->chunks = 4;
            if (h->chunks > 128) h->chunks = 128;
            if (getenv("GMPE_CHUNKS")) h->chunks = atoi(getenv("GMPE_CHUNKS"));
            if (h->chunks < 1) h->chunks = 1;
            if ((size_t)h->chunks > N) h->chunks = (int)N;
            if (getenv("GMPE_CHUNKS")) h->chunks_x = h->chunks;            // the knobs set both pipelines
            if (getenv("GMPE_AHEAD")) h->ahead_x = h->ahead;
            if ((size_t)h->chunks_x > N) h->chunks_x = (int)N;
            bool ok = true;
            for (int q2 = 0; q2 < 2 && ok; ++q2) ok = hipStreamCreateWithFlags(&h->env_st[q2], hipStreamNonBlocking) == hipSuccess;
            ok = ok && hipStreamCreateWithFlags(&h->exp_st, hipStreamNonBlocking) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) == hipSuccess;
            for (int q2 = 0; q2 < 3 && ok; ++q2) ok = hipEventCreateWithFlags(&h->ev_join[q2], hipEventDisableTiming) == hipSuccess;
            h->ev_chunk.resize((size_t)(h->chunks > h->chunks_x ? h->chunks : h->chunks_x) + 2, nullptr);
            for (size_t q2 = 0; q2 < h->ev_chunk.size() && ok; ++q2) ok = hipEventCreateWithFlags(&h->ev_chunk[q2], hipEventDisableTiming) == hipSuccess;
            h->ev_exp.resize(h->ev_chunk.size(), nullptr);
            for (size_t q2 = 0; q2 < h->ev_exp.size() && ok; ++q2) ok = hipEventCreateWithFlags(&h->ev_exp[q2], hipEventDisableTiming) == hipSuccess;
            if (!ok) { gmpe_destroy(h); return fail(GMPE_ERR_HIP, "split path: could not create the side streams / events"); }
        }
    }
    const size_t stream_f4 = (size_t)G * h->A * ((size_t)E * E / 4 + 2 * (size_t)E);
    h->block = env_block ? atoi(env_block) : (block_sel ? block_sel : (stream_f4 <= 2048 ? 64 : (stream_f4 <= 6144 ? 128 : 256)));
    if (h->block != 64 && h->block != 128 && h->block != 256) h->block = 256;
    // Multi-wave tiles specialise (wave 0: reward / info / write-back, waves 1..: graph stores). Measured with the final register
    // budgets: C2 30.2 vs 33.2 us, C4 1286 vs 1297 us, C5 shard 1901 vs 1970 us — on everywhere (GMPE_SPEC=0 turns it off).
    h->spec = getenv("GMPE_SPEC") ? atoi(getenv("GMPE_SPEC")) : 1;
    const size_t lds_step = lds_bytes(h->G, h->A, E, h->D, cfg->num_walls, 0), lds_roll = lds_bytes(h->G_roll, h->A, E, h->D, cfg->num_walls, nfuse_of(h, h->block_roll, h->ap, 2));
    const size_t lds = lds_step > lds_roll ? lds_step : lds_roll;
    if (lds > 160 * 1024) { gmpe_destroy(h); return fail(GMPE_ERR_UNSUPPORTED, "per-tile LDS exceeds 160 KiB"); }
    if (lds > 48 * 1024) {                                   // opt in to >64 KiB dynamic LDS (gfx950: 160 KiB per CU)
        const hipError_t e = sc_dispatch_lds(sc_of(h->c), (int)lds);
        if (e != hipSuccess) { gmpe_destroy(h); return fail(GMPE_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); }
    }
#ifdef GMPE_STAMPS
    {
        const size_t grid = (N + G - 1) / G;
        void* q = nullptr;
        if (hipMalloc(&q, grid * GMPE_NSTAMPS * 8) == hipSuccess) { (void)hipMemset(q, 0, grid * GMPE_NSTAMPS * 8); h->allocs.push_back(q); h->stamps = static_cast<unsigned long long*>(q); }
    }
#endif
    *out = h;
    return GMPE_OK;
}

#ifdef GMPE_STAMPS
// diagnostic build only: copy the per-workgroup phase stamps of the LAST launch to the host
int gmpe_debug_stamps(gmpe_handle* h, unsigned long long* host_dst, int64_t max_blocks) {
    const int64_t grid = (h->c.num_envs + h->G - 1) / h->G;
    const int64_t nb = grid < max_blocks ? grid : max_blocks;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(host_dst, h->stamps, (size_t)nb * GMPE_NSTAMPS * 8, hipMemcpyDeviceToHost));
    return (int)nb;
}
#endif

int gmpe_destroy(gmpe_handle* h) {
    if (h) (void)hipSetDevice(h->device);                             // streams / events / allocations below belong to the handle's device
    if (h) { for (auto& g : h->graphs) { (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); } h->graphs.clear(); if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream); h->cap_stream = nullptr; }
    if (h) {
        for (hipStream_t s : {h->env_st[0], h->env_st[1], h->exp_st}) if (s) (void)hipStreamDestroy(s);
        for (hipEvent_t e : {h->ev_fork, h->ev_join[0], h->ev_join[1], h->ev_join[2]}) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : h->ev_chunk) if (e) (void)hipEventDestroy(e);
        for (hipStream_t& s : h->part_st) if (s) { (void)hipStreamDestroy(s); s = nullptr; }
        for (hipEvent_t& e : h->part_ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
        for (hipEvent_t e : h->ev_exp) if (e) (void)hipEventDestroy(e);
        h->ev_exp.clear();
        h->ev_chunk.clear();
    }
    if (!h) return GMPE_OK;
    for (void* q : h->allocs) (void)hipFree(q);
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    if (h->edge_ws) (void)hipFree(h->edge_ws);
    if (h->edgec_ws) (void)hipFree(h->edgec_ws);
    for (hipEvent_t e : h->region_ev) if (e) (void)hipEventDestroy(e);
    delete h;
    return GMPE_OK;
}

int gmpe_set_rng_tape(gmpe_handle* h, const double* tape_dev, int64_t len_per_env) {
    if (!h) return fail(GMPE_ERR_INVALID_ARG, "null handle");
    h->s.tape = tape_dev; h->s.tape_len = tape_dev ? len_per_env : 0;
    // recorded rollouts bake the kernel parameters (tape pointer included) into their nodes: drop them
    for (auto& g : h->graphs) { (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); }
    h->graphs.clear();
    return GMPE_OK;
}

int gmpe_set_control_override(gmpe_handle* h, const double* ctrl_dev, const uint8_t* use_dev) {
    if (!h) return fail(GMPE_ERR_INVALID_ARG, "null handle");
    h->ovr = ctrl_dev; h->ovr_use = ctrl_dev ? use_dev : nullptr;
    // recorded rollouts bake the kernel parameters into their nodes: drop them
    for (auto& g : h->graphs) { (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); }
    h->graphs.clear();
    return GMPE_OK;
}
int gmpe_field_device_ptr(gmpe_handle* h, int field, void** ptr_out) {
    if (!h || !ptr_out) return fail(GMPE_ERR_INVALID_ARG, "gmpe_field_device_ptr: null argument");
    size_t b;
    return field_info(h, field, ptr_out, &b);
}

int gmpe_field_bytes(const gmpe_handle* h, int field, size_t* bytes) {
    void* p; return field_info(h, field, &p, bytes);
}
int gmpe_get_field(gmpe_handle* h, int field, void* host_dst, size_t bytes) {
    void* p; size_t b;
    int rc = field_info(h, field, &p, &b); if (rc) return rc;
    if (b != bytes) return fail(GMPE_ERR_INVALID_ARG, "gmpe_get_field: size mismatch");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipDeviceSynchronize());
    if (b) HIPCHK(hipMemcpy(host_dst, p, b, hipMemcpyDeviceToHost));
    return GMPE_OK;
}
int gmpe_set_field(gmpe_handle* h, int field, const void* host_src, size_t bytes) {
    void* p; size_t b;
    int rc = field_info(h, field, &p, &b); if (rc) return rc;
    if (b != bytes) return fail(GMPE_ERR_INVALID_ARG, "gmpe_set_field: size mismatch");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipDeviceSynchronize());
    if (b) HIPCHK(hipMemcpy(p, host_src, b, hipMemcpyHostToDevice));
    return GMPE_OK;
}

static void fill_params(const gmpe_handle* h, KParams& p, int G) {
    memset(&p, 0, sizeof p);
    p.c = h->c; p.s = h->s;
    p.A = h->A; p.L = h->L; p.O = h->O; p.E = h->E; p.D = h->D; p.F = h->F; p.G = G;
    p.ablate = h->ablate;
    p.nt = h->nt;
    p.nfuse = 0; p.rowpairs = h->rowpairs;                                // nfuse: set per launch by dispatch_env
    p.spec = h->spec;
    p.stamps = h->stamps;
    p.K = 1; p.S = 1; p.num_slots = 1;
    p.env_lo = 0; p.env_hi = h->c.num_envs;
    p.ovr = h->ovr; p.ovr_use = h->ovr_use;
    p.m_E = magic_of(p.E); p.m_AE = magic_of(p.A * p.E); p.m_EE = magic_of(p.E * p.E); p.m_nq = magic_of(p.E * p.E / 4);
    p.m_2E = magic_of(2 * p.E); p.m_pe = magic_of(p.A * p.E * 2); p.m_AD = magic_of(p.A * p.D); p.m_A = magic_of(p.A);
    p.m_L = magic_of(p.L); p.m_O = magic_of(p.O);
    p.m_C = magic_of(p.A + p.O);
    p.m_AC = magic_of(p.A * (p.A + p.O)); p.m_AEE = magic_of(p.A * p.E * p.E);
    p.m_W = magic_of(p.E * (p.E - 1) / 2); p.m_Sx = magic_of((p.E & 1) ? (p.E - 1) / 2 : p.E - 1);      // distance_pass: pairs per env, inner divisor
    p.m_FW = magic_of(p.A * (p.A - 1) / 2 + p.A * p.O);
    p.TW = gmpe_entity_table_width(&h->c); p.m_TW = magic_of(p.TW);
}
static int ap_of(const gmpe_handle* h) { return h->ap; }
static void dispatch_env(const gmpe_handle* h, int block, int ap, int fl, hipStream_t st, KParams& p) {
    p.nfuse = nfuse_of(h, block, ap, fl);
    const size_t lds = lds_bytes(p.G, h->A, h->E, h->D, h->c.num_walls, p.nfuse);
    const dim3 grid((p.env_hi - p.env_lo + p.G - 1) / p.G);
    switch (sc_of(h->c)) {
        case SC_NAV: launch_env<SC_NAV>(block, ap, fl, grid, lds, st, p); break;
        case SC_NAV_WALLS: launch_env<SC_NAV_WALLS>(block, ap, fl, grid, lds, st, p); break;
        case SC_JULY: launch_env<SC_JULY>(block, ap, fl, grid, lds, st, p); break;
        case SC_ROT: launch_env<SC_ROT>(block, ap, fl, grid, lds, st, p); break;
        case SC_TWO: launch_env<SC_TWO>(block, ap, fl, grid, lds, st, p); break;
        default: launch_env<SC_THREE>(block, ap, fl, grid, lds, st, p); break;
    }
}

// Split big-E path: one step as a software pipeline over env chunks —
//   env_st[c & 1]:  k_env(chunk c) -> compact matrices into the scratch          (latency-bound, ~10 % of HBM)
//   exp_st:         k_adj_expand(chunk c): scratch -> adj [N,A,E,E]               (HBM-bound fill, 6.5-6.9 TB/s alone)
// expand(c) waits for k_env(c) through an event, so the expansion of one chunk overlaps the fused kernel of the next; two env
// streams let one chunk's last, partly filled round of tiles overlap the next chunk's first. Fork from / join into the caller's
// stream. Run-ahead bound (`ahead`): k_env(c) waits for expand(c - ahead) so that the scratch rows the expansion re-reads A times are still in the
// Infinity Cache (gmpe_create).
// `first` / `last`: gmpe_step_many_launches chains the pipelines of consecutive open-loop steps without joining in between (fork before the first step,
// join after the last): k_env(step s+1, chunk c) then waits for expand(step s, chunk c) — the reader of the scratch rows it overwrites — instead of for
// the whole step s. (A first version with TWO scratch buffers and no bound was slower than joining per step: its live scratch did not fit the cache.)
static int split_pipeline(gmpe_handle* h, int mode, KParams& p, float* adj_full, hipStream_t st, bool first = true, bool last = true) {
    const bool chained = !(first && last);
    const uint32_t EE = (uint32_t)h->E * h->E;
    const int N = h->c.num_envs, C = chained ? h->chunks_x : h->chunks, ahead = chained ? h->ahead_x : h->ahead;
    const int per = ((N + C - 1) / C + h->G - 1) / h->G * h->G;              // envs per chunk, a whole number of tiles
    const int fl = (mode == MODE_STEP && !h->nt && !h->ablate && h->spec) ? 1 : 0;
    p.o.adj = h->adj_scratch; p.o.adj_compact = 1;
    if (first) {
        HIPCHK(hipEventRecord(h->ev_fork, st));
        for (hipStream_t s : {h->env_st[0], h->env_st[1], h->exp_st}) HIPCHK(hipStreamWaitEvent(s, h->ev_fork, 0));
    }
    // chunk boundaries: equal chunks, except that the first two are a quarter and a half chunk (GMPE_RAMP) so that the expansion stream — the
    // bottleneck — starts after a quarter of a k_env chunk instead of a whole one. Chained steps: only for pipelines of >= 6 chunks per step (us per
    // step with / without: 1024 envs 729 / 697, 2048: 1375-1431 / 1350, 4096: 2750 / 2790, 8192: 5818 / 6000, 16384: 10941-11219 / 11327-11368)
    int bounds[134]; int nb = 0;                                            // <= 128 chunks + the two ramp chunks + the end marker
    {
        const bool ramp = (h->ramp >= 0 ? h->ramp != 0 : (!chained || C >= 6)) && per >= 4 * h->G;   // GMPE_RAMP=0: equal chunks
        int lo = 0;
        if (ramp) { const int q = (per / 4 + h->G - 1) / h->G * h->G; bounds[nb++] = lo; lo += q; bounds[nb++] = lo; lo += 2 * q; }
        while (lo < N && nb < 132) { bounds[nb++] = lo; lo += per; }
        bounds[nb] = N;
        for (int q = 0; q < nb; ++q) if (bounds[q] > N) bounds[q] = N;
        // every env must be covered by a chunk that has its events: never step a truncated batch silently
        if (lo < N || nb > (int)h->ev_chunk.size() || nb > (int)h->ev_exp.size())
            return fail(GMPE_ERR_INVALID_ARG, "split pipeline: " + std::to_string(nb) + " chunks of " + std::to_string(per) + " envs do not fit the " +
                                              std::to_string(h->ev_chunk.size()) + " chunk events sized by gmpe_create (GMPE_CHUNKS / GMPE_RAMP changed after create?)");
    }
    for (int c = 0; c < nb; ++c) {
        const int lo = bounds[c], hi = bounds[c + 1];
        if (lo >= hi) continue;
        hipStream_t se = h->env_st[c & 1];
        p.env_lo = lo; p.env_hi = hi;
        if (ahead > 0 && c >= ahead) HIPCHK(hipStreamWaitEvent(se, h->ev_exp[c - ahead], 0));
        if (chained && !first) HIPCHK(hipStreamWaitEvent(se, h->ev_exp[c], 0));   // the previous step's expansion of this chunk has read the scratch rows
        dispatch_env(h, h->block, ap_of(h), fl, se, p);
        HIPCHK(hipEventRecord(h->ev_chunk[c], se));
        HIPCHK(hipStreamWaitEvent(h->exp_st, h->ev_chunk[c], 0));
        const int gy = hi - lo < 1024 ? hi - lo : 1024;
        if ((EE & 3) == 0) hipLaunchKernelGGL(k_adj_expand, dim3(((uint32_t)h->A * (EE / 4) + 255) / 256, gy), dim3(256), 0, h->exp_st,
                                              h->adj_scratch, adj_full, lo, hi, EE / 4, h->A, magic_of(EE / 4));
        else hipLaunchKernelGGL(k_adj_expand1, dim3(((uint32_t)h->A * EE + 255) / 256, gy), dim3(256), 0, h->exp_st, h->adj_scratch, adj_full, lo, hi, EE, h->A);
        if (ahead > 0 || chained) HIPCHK(hipEventRecord(h->ev_exp[c], h->exp_st));
    }
    if (last) {
        int q = 0;
        for (hipStream_t s : {h->env_st[0], h->env_st[1], h->exp_st}) { HIPCHK(hipEventRecord(h->ev_join[q], s)); HIPCHK(hipStreamWaitEvent(st, h->ev_join[q], 0)); ++q; }
    }
    HIPCHK(hipGetLastError());
    return GMPE_OK;
}

static int launch(gmpe_handle* h, int mode, const int32_t* act, const float* onehot, const uint8_t* mask,
                  const gmpe_outputs* out, void* stream, int env_lo = 0, int env_hi = -1) {
    if (!h) return fail(GMPE_ERR_INVALID_ARG, "null handle");
    KParams p;
    fill_params(h, p, h->G);
    if (out) p.o = *out;
    p.act = act; p.onehot = onehot; p.mask = mask; p.mode = mode;
    if (env_hi >= 0) { p.env_lo = env_lo; p.env_hi = env_hi; }           // gmpe_step_envs: a contiguous env range (pointers stay whole-batch)
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->timing && !h->capturing) {
        if (h->ev_used + 2 > h->ev.size()) {
            for (int q = 0; q < 2; ++q) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); h->ev.push_back(e); }
        }
        e0 = h->ev[h->ev_used]; e1 = h->ev[h->ev_used + 1];
        HIPCHK(hipEventRecord(e0, st));
    }
    if (h->split && p.o.adj && !p.o.adj_compact) {
        // the fused kernel writes the env's single matrix into the scratch, k_adj_expand makes the A copies (chunk pipeline)
        float* adj_full = p.o.adj;
        const int rc = split_pipeline(h, mode, p, adj_full, st);
        if (rc) return rc;
    } else {
        const int fl = (mode == MODE_STEP && !h->nt && !h->ablate && h->spec) ? 1 : 0;   // steady-state instantiation (flags folded)
        dispatch_env(h, h->block, ap_of(h), fl, st, p);
    }
    HIPCHK(hipGetLastError());
    if (h->timing && !h->capturing) { HIPCHK(hipEventRecord(e1, st)); h->ev_used += 2; }
    return GMPE_OK;
}

int gmpe_rollout_steps(gmpe_handle* h, const int32_t* actions_dev, const gmpe_rollout* r, const gmpe_outputs* slot0, void* stream) {
    if (!h || !actions_dev || !r) return fail(GMPE_ERR_INVALID_ARG, "gmpe_rollout_steps: null argument");
    if (r->num_steps < 1 || r->num_action_sets < 1 || r->num_slots < 1 || r->first_slot < 0 || r->first_slot >= r->num_slots)
        return fail(GMPE_ERR_INVALID_ARG, "gmpe_rollout_steps: bad step / slot counts");
    if (h->split) return fail(GMPE_ERR_UNSUPPORTED, "gmpe_rollout_steps: this handle runs the split big-E path (use gmpe_step_many)");

    KParams p;
    fill_params(h, p, h->G_roll);
    if (slot0) p.o = *slot0;
    p.act = actions_dev; p.mode = MODE_STEP;
    p.K = r->num_steps; p.S = r->num_action_sets; p.num_slots = r->num_slots; p.first_slot = r->first_slot;
    p.st_obs = r->stride_obs; p.st_id = r->stride_agent_id; p.st_node = r->stride_node_obs; p.st_adj = r->stride_adj;
    p.st_rew = r->stride_reward; p.st_done = r->stride_done; p.st_info = r->stride_info; p.st_mask = r->stride_masks; p.st_tab = r->stride_entity_table;
    p.masks = r->masks; p.active = r->active_masks;
    {   // a rollout that fills more slots than the 256 MiB Infinity Cache holds streams past it: nontemporal graph stores (like the big launches)
        const double step_bytes = (double)h->c.num_envs * h->A * ((double)h->E * h->E * (p.o.adj_compact ? 1.0 / h->A : 1.0) + (double)h->F * h->E) * 4.0;
        p.nt = h->rollnt >= 0 ? h->rollnt : (step_bytes * r->num_slots > 192.0 * 1024 * 1024 ? 1 : 0);      // GMPE_ROLLNT was read by gmpe_create: no getenv on the collect path
    }
    HIPCHK(hipSetDevice(h->device));
    dispatch_env(h, h->block_roll, ap_of(h), 2, static_cast<hipStream_t>(stream), p);
    HIPCHK(hipGetLastError());
    return GMPE_OK;
}

int gmpe_get_tuning(const gmpe_handle* h, gmpe_tuning* t) {
    if (!h || !t) return fail(GMPE_ERR_INVALID_ARG, "gmpe_get_tuning: null argument");
    memset(t, 0, sizeof *t);
    t->G = h->G; t->block = h->block; t->nt = h->nt; t->spec = h->spec; t->split = h->split; t->roll = h->roll; t->ap = ap_of(h);
    t->lds_bytes = (int32_t)lds_bytes(h->G, h->A, h->E, h->D, h->c.num_walls, 0);
    t->lds_bytes_roll = (int32_t)lds_bytes(h->G_roll, h->A, h->E, h->D, h->c.num_walls, nfuse_of(h, h->block_roll, h->ap, 2));
    t->G_roll = h->G_roll; t->block_roll = h->block_roll;
    t->chunks = h->split ? h->chunks : 0; t->ahead = h->split ? h->ahead : 0;
    t->xstep = h->split ? h->xstep : 0; t->chunks_x = h->split ? h->chunks_x : 0; t->ahead_x = h->split ? h->ahead_x : 0;
#ifdef GMPE_DIAG
    t->diag_build = 1;
#endif
    return GMPE_OK;
}

int gmpe_reset(gmpe_handle* h, const uint8_t* env_mask_dev, const gmpe_outputs* out, void* stream) {
    return launch(h, MODE_RESET, nullptr, nullptr, env_mask_dev, out, stream);
}
int gmpe_step(gmpe_handle* h, const int32_t* action_idx_dev, const gmpe_outputs* out, void* stream) {
    if (!action_idx_dev) return fail(GMPE_ERR_INVALID_ARG, "gmpe_step: null actions");
    return launch(h, MODE_STEP, action_idx_dev, nullptr, nullptr, out, stream);
}
int gmpe_step_envs(gmpe_handle* h, const int32_t* action_idx_dev, const gmpe_outputs* out, int32_t env_lo, int32_t env_hi, void* stream) {
    if (!h || !action_idx_dev) return fail(GMPE_ERR_INVALID_ARG, "gmpe_step_envs: null argument");
    if (env_lo < 0 || env_hi > h->c.num_envs || env_lo >= env_hi) return fail(GMPE_ERR_INVALID_ARG, "gmpe_step_envs: bad env range");
    if (h->split && out && out->adj && !out->adj_compact) return fail(GMPE_ERR_UNSUPPORTED, "gmpe_step_envs: this handle runs the split big-E path (whole-batch steps only)");
    return launch(h, MODE_STEP, action_idx_dev, nullptr, nullptr, out, stream, env_lo, env_hi);
}
int gmpe_step_many_envs(gmpe_handle* h, const int32_t* actions_dev, int32_t num_steps, int32_t num_action_sets, const gmpe_outputs* out,
                        int32_t parts, void* stream) {
    if (!h || !actions_dev || num_steps < 0 || num_action_sets < 1) return fail(GMPE_ERR_INVALID_ARG, "gmpe_step_many_envs: bad arguments");
    if (parts < 1 || parts > 4 || parts > h->c.num_envs) return fail(GMPE_ERR_INVALID_ARG, "gmpe_step_many_envs: 1..4 env ranges");
    if (h->split && out && out->adj && !out->adj_compact) return fail(GMPE_ERR_UNSUPPORTED, "gmpe_step_many_envs: split big-E path");
    if (num_steps == 0) return GMPE_OK;
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int q = 0; q < parts; ++q) if (!h->part_st[q]) HIPCHK(hipStreamCreateWithFlags(&h->part_st[q], hipStreamNonBlocking));
    for (int q = 0; q <= parts; ++q) if (!h->part_ev[q]) HIPCHK(hipEventCreateWithFlags(&h->part_ev[q], hipEventDisableTiming));
    HIPCHK(hipEventRecord(h->part_ev[parts], st));
    const size_t stride = (size_t)h->c.num_envs * h->A;
    const int N = h->c.num_envs;
    for (int q = 0; q < parts; ++q) HIPCHK(hipStreamWaitEvent(h->part_st[q], h->part_ev[parts], 0));
    for (int32_t k = 0; k < num_steps; ++k)
        for (int q = 0; q < parts; ++q) {
            // range q's step k depends only on range q's step k-1 (same side stream): its latency chain runs under the other ranges' store drains
            const int lo = (int)((long long)N * q / parts / h->G * h->G), hi = q + 1 == parts ? N : (int)((long long)N * (q + 1) / parts / h->G * h->G);
            if (lo >= hi) continue;
            const int rc = launch(h, MODE_STEP, actions_dev + (size_t)(k % num_action_sets) * stride, nullptr, nullptr, out, h->part_st[q], lo, hi);
            if (rc) {                                                   // join what is already enqueued, so the caller's stream stays ordered after it
                const std::string msg = g_err;
                for (int j = 0; j < parts; ++j) if (hipEventRecord(h->part_ev[j], h->part_st[j]) == hipSuccess) (void)hipStreamWaitEvent(st, h->part_ev[j], 0);
                return fail(rc, msg);
            }
        }
    for (int q = 0; q < parts; ++q) { HIPCHK(hipEventRecord(h->part_ev[q], h->part_st[q])); HIPCHK(hipStreamWaitEvent(st, h->part_ev[q], 0)); }
    return GMPE_OK;
}
static bool same_out(const gmpe_outputs& a, const gmpe_outputs& b) {
    return a.obs == b.obs && a.agent_id == b.agent_id && a.node_obs == b.node_obs && a.adj == b.adj && a.reward == b.reward &&
           a.done == b.done && a.info == b.info && a.adj_compact == b.adj_compact && a.entity_table == b.entity_table;
}
static int find_graph(const gmpe_handle* h, const int32_t* actions_dev, int32_t K, int32_t S, const gmpe_outputs& o) {
    for (size_t q = 0; q < h->graphs.size(); ++q) {
        const auto& g = h->graphs[q];
        if (g.actions == actions_dev && g.K == K && g.S == S && same_out(g.out, o)) return (int)q;
    }
    return -1;
}
int gmpe_step_many_prepare(gmpe_handle* h, const int32_t* actions_dev, int32_t num_steps, int32_t num_action_sets, const gmpe_outputs* out) {
    if (!h || !actions_dev || num_steps < 1 || num_action_sets < 1) return fail(GMPE_ERR_INVALID_ARG, "gmpe_step_many_prepare: bad arguments");
    gmpe_outputs o; memset(&o, 0, sizeof o); if (out) o = *out;
    if (find_graph(h, actions_dev, num_steps, num_action_sets, o) >= 0) return GMPE_OK;
    if (h->split) return GMPE_OK;                                 // the split path forks onto side streams per step: not recorded, gmpe_step_many loops
    HIPCHK(hipSetDevice(h->device));
    if (!h->cap_stream) HIPCHK(hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
    if (h->graphs.size() >= 4) {                                  // small cache: drop the oldest
        (void)hipGraphExecDestroy(h->graphs[0].exec); (void)hipGraphDestroy(h->graphs[0].graph);
        h->graphs.erase(h->graphs.begin());
    }
    gmpe_handle::StepGraph g; g.actions = actions_dev; g.K = num_steps; g.S = num_action_sets; g.out = o; g.graph = nullptr; g.exec = nullptr;
    HIPCHK(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
    h->capturing = true;
    const size_t stride = (size_t)h->c.num_envs * h->A;
    int rc = GMPE_OK;
    for (int32_t k = 0; k < num_steps && rc == GMPE_OK; ++k)
        rc = launch(h, MODE_STEP, actions_dev + (size_t)(k % num_action_sets) * stride, nullptr, nullptr, out, h->cap_stream);
    h->capturing = false;
    const hipError_t e = hipStreamEndCapture(h->cap_stream, &g.graph);
    if (rc) { if (g.graph) (void)hipGraphDestroy(g.graph); return rc; }
    if (e != hipSuccess) return fail(GMPE_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    const hipError_t e2 = hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0);
    if (e2 != hipSuccess) { (void)hipGraphDestroy(g.graph); return fail(GMPE_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e2)); }
    h->graphs.push_back(g);
    return GMPE_OK;
}
int gmpe_step_many(gmpe_handle* h, const int32_t* actions_dev, int32_t num_steps, int32_t num_action_sets,
                   const gmpe_outputs* out, void* stream) {
    if (!h || !actions_dev || num_steps < 0 || num_action_sets < 1) return fail(GMPE_ERR_INVALID_ARG, "gmpe_step_many: bad arguments");
    if (num_steps == 0) return GMPE_OK;
    if (h->roll && !h->split && !h->timing && !h->ablate) {                  // one launch: the persistent rollout kernel (same results as the launch loop)
        gmpe_rollout r; memset(&r, 0, sizeof r);
        r.num_steps = num_steps; r.num_action_sets = num_action_sets; r.num_slots = 1;
        return gmpe_rollout_steps(h, actions_dev, &r, out, stream);
    }
    return gmpe_step_many_launches(h, actions_dev, num_steps, num_action_sets, out, stream);
}
int gmpe_step_many_launches(gmpe_handle* h, const int32_t* actions_dev, int32_t num_steps, int32_t num_action_sets,
                            const gmpe_outputs* out, void* stream) {
    if (!h || !actions_dev || num_steps < 0 || num_action_sets < 1) return fail(GMPE_ERR_INVALID_ARG, "gmpe_step_many_launches: bad arguments");
    if (num_steps == 0) return GMPE_OK;
    if (!h->timing) {                                             // per-launch event pairs need individual launches
        gmpe_outputs o; memset(&o, 0, sizeof o); if (out) o = *out;
        const int q = find_graph(h, actions_dev, num_steps, num_action_sets, o);
        if (q >= 0) {                                             // prepared: one graph launch replays the K kernel nodes
            HIPCHK(hipSetDevice(h->device));
            HIPCHK(hipGraphLaunch(h->graphs[q].exec, static_cast<hipStream_t>(stream)));
            return GMPE_OK;
        }
    }
    const size_t stride = (size_t)h->c.num_envs * h->A;
    if (h->split && h->xstep && !h->timing && num_steps > 1 && out && out->adj && !out->adj_compact) {
        // open-loop steps on the split path: one chunk pipeline over all the steps (no join between steps)
        HIPCHK(hipSetDevice(h->device));
        for (int32_t k = 0; k < num_steps; ++k) {
            KParams p;
            fill_params(h, p, h->G);
            p.o = *out; p.act = actions_dev + (size_t)(k % num_action_sets) * stride; p.mode = MODE_STEP;
            const int rc = split_pipeline(h, MODE_STEP, p, out->adj, static_cast<hipStream_t>(stream), k == 0, k == num_steps - 1);
            if (rc) return rc;
        }
        return GMPE_OK;
    }
    for (int32_t k = 0; k < num_steps; ++k) {
        const int rc = launch(h, MODE_STEP, actions_dev + (size_t)(k % num_action_sets) * stride, nullptr, nullptr, out, stream);
        if (rc) return rc;
    }
    return GMPE_OK;
}
int gmpe_step_onehot(gmpe_handle* h, const float* onehot_dev, const gmpe_outputs* out, void* stream) {
    if (!onehot_dev) return fail(GMPE_ERR_INVALID_ARG, "gmpe_step_onehot: null actions");
    return launch(h, MODE_STEP, nullptr, onehot_dev, nullptr, out, stream);
}


int gmpe_edges_from_adj(gmpe_handle* h, const float* adj_dev, int32_t batch, int32_t num_nodes, float max_edge_dist,
                        int32_t inclusive, int32_t* edge_index_dev, float* edge_attr_dev, int32_t cap,
                        int32_t* n_edges_dev, void* stream) {
    if (!h || !adj_dev || !edge_index_dev || !edge_attr_dev || !n_edges_dev) return fail(GMPE_ERR_INVALID_ARG, "gmpe_edges_from_adj: null argument");
    if (batch < 1 || num_nodes < 1 || cap < 0) return fail(GMPE_ERR_INVALID_ARG, "gmpe_edges_from_adj: bad sizes");
    if ((int64_t)batch * num_nodes > INT32_MAX) return fail(GMPE_ERR_INVALID_ARG, "gmpe_edges_from_adj: node ids overflow int32");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (h->edge_ws_graphs < (size_t)batch) {                 // workspace grows on first use only
        if (h->edge_ws) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipFree(h->edge_ws)); h->edge_ws = nullptr; }
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->edge_ws), sizeof(int32_t) * (2 * (size_t)batch + (size_t)batch / 1024 + 2)));
        h->edge_ws_graphs = batch;
    }
    int32_t* counts = h->edge_ws; int32_t* offsets = h->edge_ws + h->edge_ws_graphs;
    const int EE = num_nodes * num_nodes, nchunks = (batch + 1023) / 1024;
    hipLaunchKernelGGL(k_edge_count, dim3((batch + 3) / 4), dim3(256), 0, st, adj_dev, batch, EE, max_edge_dist, inclusive, counts);
    hipLaunchKernelGGL(k_edge_scan, dim3(nchunks), dim3(1024), 0, st, counts, batch, offsets, n_edges_dev);
    hipLaunchKernelGGL(k_edge_write, dim3((batch + 3) / 4), dim3(256), 0, st, adj_dev, batch, num_nodes, max_edge_dist, inclusive, offsets,
                       edge_index_dev, edge_attr_dev, cap);
    HIPCHK(hipGetLastError());
    return GMPE_OK;
}

int gmpe_edges_from_adj_compact(gmpe_handle* h, const float* adj_compact_dev, int32_t num_envs, int32_t copies, int32_t num_nodes,
                                float max_edge_dist, int32_t inclusive, int32_t index64, void* edge_index_dev, float* edge_attr_dev,
                                int64_t cap, int32_t* n_edges_dev, void* stream) {
    if (!h || !adj_compact_dev || !edge_index_dev || !edge_attr_dev || !n_edges_dev) return fail(GMPE_ERR_INVALID_ARG, "gmpe_edges_from_adj_compact: null argument");
    if (num_envs < 1 || copies < 1 || num_nodes < 1 || cap < 0) return fail(GMPE_ERR_INVALID_ARG, "gmpe_edges_from_adj_compact: bad sizes");
    if (!index64 && (int64_t)num_envs * copies * num_nodes > INT32_MAX) return fail(GMPE_ERR_INVALID_ARG, "gmpe_edges_from_adj_compact: node ids overflow int32 (use index64)");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t need = (size_t)num_envs + (size_t)(num_envs + 3) / 4 + 2;
    if (h->edgec_ws_n < need) {                              // workspace grows on first use only
        if (h->edgec_ws) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipFree(h->edgec_ws)); h->edgec_ws = nullptr; }
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->edgec_ws), sizeof(int32_t) * need));
        h->edgec_ws_n = need;
    }
    int32_t* counts = h->edgec_ws; int32_t* btot = h->edgec_ws + num_envs;
    const int nb = (num_envs + 3) / 4, EE = num_nodes * num_nodes;
    hipLaunchKernelGGL(k_edgec_count, dim3(nb), dim3(256), 0, st, adj_compact_dev, num_envs, EE, max_edge_dist, inclusive, counts, btot);
    if (index64) hipLaunchKernelGGL((k_edgec_write<true>), dim3(nb), dim3(256), 0, st, adj_compact_dev, num_envs, copies, num_nodes, max_edge_dist, inclusive,
                                    counts, btot, edge_index_dev, edge_attr_dev, (long long)cap, n_edges_dev);
    else hipLaunchKernelGGL((k_edgec_write<false>), dim3(nb), dim3(256), 0, st, adj_compact_dev, num_envs, copies, num_nodes, max_edge_dist, inclusive,
                            counts, btot, edge_index_dev, edge_attr_dev, (long long)cap, n_edges_dev);
    HIPCHK(hipGetLastError());
    return GMPE_OK;
}

int gmpe_masks_from_dones(gmpe_handle* h, const uint8_t* done_dev, float* masks_dev, float* active_masks_dev, void* stream) {
    if (!h || !done_dev) return fail(GMPE_ERR_INVALID_ARG, "gmpe_masks_from_dones: bad arguments");
    HIPCHK(hipSetDevice(h->device));
    const int total = h->c.num_envs * h->A;
    hipLaunchKernelGGL(k_masks, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), done_dev, h->c.num_envs, h->A, masks_dev, active_masks_dev);
    HIPCHK(hipGetLastError());
    return GMPE_OK;
}
int gmpe_timing_mark(gmpe_handle* h, int32_t which, void* stream) {
    if (!h || which < 0 || which > 1) return fail(GMPE_ERR_INVALID_ARG, "gmpe_timing_mark: bad arguments");
    HIPCHK(hipSetDevice(h->device));
    if (!h->region_ev[which]) HIPCHK(hipEventCreate(&h->region_ev[which]));
    HIPCHK(hipEventRecord(h->region_ev[which], static_cast<hipStream_t>(stream)));
    return GMPE_OK;
}
int gmpe_timing_region_ms(gmpe_handle* h, double* ms) {
    if (!h || !ms || !h->region_ev[0] || !h->region_ev[1]) return fail(GMPE_ERR_INVALID_ARG, "gmpe_timing_region_ms: region not marked");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipEventSynchronize(h->region_ev[1]));
    float f = 0;
    HIPCHK(hipEventElapsedTime(&f, h->region_ev[0], h->region_ev[1]));
    *ms = f;
    return GMPE_OK;
}
int gmpe_timing_enable(gmpe_handle* h, int32_t enable) {
    if (!h) return fail(GMPE_ERR_INVALID_ARG, "null handle");
    h->timing = enable != 0;
    return GMPE_OK;
}
int gmpe_timing_read(gmpe_handle* h, double* total_ms, int64_t* launches, int32_t reset_counters) {
    if (!h) return fail(GMPE_ERR_INVALID_ARG, "null handle");
    HIPCHK(hipSetDevice(h->device));
    for (size_t q = 0; q + 1 < h->ev_used; q += 2) {
        HIPCHK(hipEventSynchronize(h->ev[q + 1]));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, h->ev[q], h->ev[q + 1]));
        h->t_total_ms += ms; h->t_launches += 1;
    }
    h->ev_used = 0;
    if (total_ms) *total_ms = h->t_total_ms;
    if (launches) *launches = h->t_launches;
    if (reset_counters) { h->t_total_ms = 0; h->t_launches = 0; }
    return GMPE_OK;
}

}  // extern "C"
