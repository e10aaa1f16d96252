// np.random.random(n) on the device, bit for bit (the stochastic-rounding draws of
// _static_quantize_padding_asymmetric, jzf_quantize.py:55-67, come from NumPy's global MT19937 generator).
//
// The reference's quantiser is bit-exact only with NumPy's own stream, so round 2 drew the uniforms on the host and shipped them:
// 8 bytes per element over PCIe -- more than the fp32 input -- after ~3 ns per draw on one host core (27-35 ms per 1e7 elements, 15x
// the whole cipher round).  MT19937 is a 19937-bit linear recurrence: word k of the next 624-word block depends on words k, k + 1
// and k + 397 of the current one, so a block is three dependent phases of 227 + 227 + 170 independent words and blocks are strictly
// sequential: one wave walks a stream at ~0.5 us per block (0.64 G draws/s, the first version of this file).  But the recurrence is
// LINEAR over GF(2), so the state J words ahead is a fixed GF(2) combination of the next 19937 + 623 words: the stream is cut into
// substreams of 65,536 doubles, their starting states are found by jumping ahead (a binary tree of jumps by 2^j substreams, the
// jump polynomials x^(S 2^j) mod phi built once per process on the host, each jump a convolution over freshly generated words), and
// all substreams then run at once, one workgroup each: inside it ONE wave runs the twist chain (three LDS phases per block,
// wave-synchronous, no workgroup barrier on the critical path) eight blocks ahead into a ring while the other fifteen waves temper the
// finished blocks and emit doubles exactly as NumPy's mt19937_next_double does (a = next >> 5, b = next >> 6, (a * 2^26 + b) / 2^53).
// 1e7 doubles: 0.5 ms (20 G draws/s) against 27-35 ms for np.random.random.  The state the caller passes in (np.random.get_state())
// is advanced exactly as NumPy would have advanced it, so host draws continue the stream.
#include "ctx.h"

#include <algorithm>
#include <cstring>
#include <mutex>
#include <vector>

using flashe_host::fail;

namespace {

constexpr int kMtN = 624, kMtM = 397, kMtThreads = 1024;

__device__ __forceinline__ uint32_t mt_temper(uint32_t y)
{
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

__device__ __forceinline__ uint32_t mt_next(uint32_t cur, uint32_t nxt, uint32_t far)
{
    const uint32_t y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__device__ __forceinline__ double mt_double(uint32_t a, uint32_t b)
{
    return (static_cast<double>(a >> 5) * 67108864.0 + static_cast<double>(b >> 6)) / 9007199254740992.0;
}

// One 624-word block from the previous one, by ONE wave (no workgroup barrier on the critical path): three dependent phases of at
// most four words per lane; LDS operations of one wave execute in order, the fences only stop the compiler from reordering them.
__device__ __forceinline__ void mt_twist_wave(const uint32_t *__restrict__ o, uint32_t *__restrict__ nw, int lane)
{
    constexpr int D = kMtN - kMtM;                   // 227
    // a lane owns FOUR CONSECUTIVE words of a phase, so that its reads and writes merge into 8- and 16-byte LDS accesses (a lone
    // wave issues LDS instructions slowly: the instruction count, not the bytes, is what a phase costs)
    {
        const int k = 4 * lane;                      // 0 .. 224 (lane 56 owns 224, 225, 226)
        if (k < D) {
            uint32_t c[5], f[4], r[4];
#pragma unroll
            for (int j = 0; j < 5; j++) c[j] = o[k + j];
#pragma unroll
            for (int j = 0; j < 4; j++) f[j] = o[k + kMtM + j < kMtN ? k + kMtM + j : kMtN - 1];
#pragma unroll
            for (int j = 0; j < 4; j++) r[j] = mt_next(c[j], c[j + 1], f[j]);
#pragma unroll
            for (int j = 0; j < 4; j++) if (k + j < D) nw[k + j] = r[j];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
        const int k = D + 4 * lane;                  // 227 .. 451 (lane 56 owns 451, 452, 453)
        if (k < 2 * D) {
            uint32_t c[5], f[4], r[4];
#pragma unroll
            for (int j = 0; j < 5; j++) c[j] = o[k + j];
#pragma unroll
            for (int j = 0; j < 4; j++) f[j] = nw[k - D + j];
#pragma unroll
            for (int j = 0; j < 4; j++) r[j] = mt_next(c[j], c[j + 1], f[j]);
#pragma unroll
            for (int j = 0; j < 4; j++) if (k + j < 2 * D) nw[k + j] = r[j];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
        const int k = 2 * D + 4 * lane;              // 454 .. 622 (lane 42 owns 622, 623)
        if (k < kMtN) {
            uint32_t c[5], f[4], r[4];
#pragma unroll
            for (int j = 0; j < 5; j++) c[j] = k + j < kMtN ? o[k + j] : nw[0];          // word 624 of the old block = word 0 of the NEW one
#pragma unroll
            for (int j = 0; j < 4; j++) f[j] = nw[k - D + j < kMtN ? k - D + j : kMtN - 1];
#pragma unroll
            for (int j = 0; j < 4; j++) r[j] = mt_next(c[j], c[j + 1], f[j]);
#pragma unroll
            for (int j = 0; j < 4; j++) if (k + j < kMtN) nw[k + j] = r[j];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- the stream, cut into substreams that run in parallel ---------------------------------------------------------------------
// Word positions: x[0..623] is the key as passed in, x[624 + i] = f(x[i], x[i + 1], x[i + 397]); double i uses positions pos0 + 2 i
// and pos0 + 2 i + 1.  Substream 0 starts from the key and emits the doubles whose FIRST word lies in [pos0, S + 1); substream k >= 1
// starts from the WINDOW V_k = x[k S + 1 .. k S + 624] -- 624 consecutive raw words are a complete generator state, the recurrence
// is shift invariant -- and emits the doubles whose first word lies in [k S + 1, (k + 1) S + 1) (it generates one word beyond its
// range for a pair that straddles the boundary).  Inside a workgroup: wave 0 runs the twist chain kMtBatch blocks ahead into one
// half of a two-batch ring while waves 1..15 temper the other half and emit its doubles (the first word of a pair that straddles
// two batches travels through `carry`).  The words of the block NumPy's state ends in are written out raw.
constexpr int kMtBatch = 8;

struct MtGen {
    const uint32_t *key;        // 624 words: the state as passed in
    const uint32_t *windows;    // P x 624 words (entry 0 unused)
    uint64_t pos0, n, S;
    uint32_t P;
    uint64_t final_block;       // index t of the 624-aligned block the generator state ends in
    uint32_t *state_out;        // its 624 raw words
};

__global__ __launch_bounds__(kMtThreads) void mt19937_generate_kernel(const MtGen a, double *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) uint32_t ring[2][kMtBatch][kMtN];
    __shared__ uint32_t carry[2];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const uint64_t k = blockIdx.x, pos0 = a.pos0, end = pos0 + 2 * a.n;
    const bool last = k + 1 == a.P;
    const uint64_t w0 = k ? k * a.S + 1 : 0;                               // position of the first word of block 0 of this substream
    const uint64_t lo = k ? w0 : pos0, hi = last ? ~0ull : (k + 1) * a.S + 1;   // first words this substream owns
    const uint64_t fin0 = a.final_block * kMtN;
    const uint64_t gen_end = last ? (end > fin0 + kMtN ? end : fin0 + kMtN) : hi + 1;      // one past the last word it must produce
    const uint64_t n_blocks = (gen_end - w0 + kMtN - 1) / kMtN;
    const uint64_t n_batches = (n_blocks + kMtBatch - 1) / kMtBatch;
    constexpr uint64_t kBW = static_cast<uint64_t>(kMtBatch) * kMtN;
    const uint32_t *src0 = k ? a.windows + k * kMtN : a.key;

    auto produce = [&](uint64_t g) {                                 // wave 0: the blocks of batch g
        for (int j = 0; j < kMtBatch; j++) {
            const uint64_t bi = g * kMtBatch + j;
            if (bi >= n_blocks) break;
            uint32_t *dst = ring[g & 1][j];
            if (bi == 0) {
                for (int i = lane; i < kMtN; i += 64) dst[i] = src0[i];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            } else {
                const uint32_t *src = j ? ring[g & 1][j - 1] : ring[(g - 1) & 1][kMtBatch - 1];
                mt_twist_wave(src, dst, lane);
            }
        }
    };
    auto consume = [&](uint64_t g) {                                 // waves 1..15: the doubles whose SECOND word lies in batch g
        const uint64_t base = w0 + g * kBW;
        const uint32_t(*blk)[kMtN] = ring[g & 1];
        constexpr int kConsumers = kMtThreads - 64, kPairs = kMtN / 2;              // at most 312 second words per block
        for (int u = t - 64; u < kMtBatch * kPairs; u += kConsumers) {
            const int j = u / kPairs, q = u - j * kPairs;                           // block of the batch, pair slot of the block
            const uint64_t b0 = base + static_cast<uint64_t>(j) * kMtN;             // position of the block's word 0
            if (b0 >= gen_end) break;
            // second words sit at the block offsets x with (b0 + x - pos0) odd
            const uint32_t x = 2u * q + (((b0 - pos0) & 1u) ? 0u : 1u);
            const uint64_t P2 = b0 + x;
            if (x >= kMtN || P2 >= end || P2 <= lo || P2 - 1 >= hi) continue;       // first word P2 - 1 in [lo, hi), both words used
            const uint32_t wb = mt_temper(blk[j][x]);
            const uint32_t wa = x ? mt_temper(blk[j][x - 1]) : j ? mt_temper(blk[j - 1][kMtN - 1]) : carry[g & 1];
            out[(P2 - pos0 - 1) >> 1] = mt_double(wa, wb);
        }
        // the batch ends with the FIRST word of a pair: hand it to the next batch
        if (t == 64 && ((base + kBW - 1 - pos0) & 1u) == 0) carry[(g + 1) & 1] = mt_temper(blk[kMtBatch - 1][kMtN - 1]);
        // raw words of the final block
        for (int j = 0; j < kMtBatch; j++) {
            const uint64_t b0 = base + static_cast<uint64_t>(j) * kMtN;
            if (b0 >= gen_end) break;
            if (b0 + kMtN <= fin0 || b0 >= fin0 + kMtN) continue;                   // (wave-uniform)
            for (int i = t - 64; i < kMtN; i += kConsumers) {
                const uint64_t P = b0 + i;
                // (every position of the final block is written by exactly one substream: substream 0 also owns the words below pos0)
                if (P >= fin0 && P < fin0 + kMtN && (k == 0 || P >= lo) && (P < hi || last)) a.state_out[P - fin0] = blk[j][i];
            }
        }
    };

    if (wave == 0) produce(0);
    __syncthreads();
    for (uint64_t g = 0; g < n_batches; g++) {
        if (wave == 0) { if (g + 1 < n_batches) produce(g + 1); }
        else consume(g);
        __syncthreads();
    }
}

// ---- where the substreams start: jumping ahead --------------------------------------------------------------------------------
// y_i = x[i + 1] is a coordinate sequence of the 19937-bit generator state, so it obeys the linear recurrence given by the
// characteristic polynomial phi of the state transition (135 terms, degree 19937): sum_k phi_k y[i + k] = 0.  Hence, with
// g = x^J mod phi:  y[J + w] = sum_k g_k y[k + w]  -- a window J words ahead is a GF(2) combination of the 19937 + 623 words that
// follow the current window (Haramoto et al.'s jump-ahead, evaluated as a convolution instead of by Horner: the words are produced
// by the ordinary recurrence -- 33 blocks -- and every output word is an independent XOR).  The windows V_k are filled level by
// level: V_{a + d} from V_a with g = x^(d S) for d = 2^(L-1), ..., 2, 1, all jumps of a level in parallel, each jump split over M
// workgroups that take a slice of the coefficients and XOR their partial windows into global memory.
constexpr int kJumpSeqBlocks = 33, kJumpSeq = kJumpSeqBlocks * kMtN;       // 20,592 >= 19,937 + 623

__global__ __launch_bounds__(kMtThreads) void mt19937_jump_kernel(uint32_t *__restrict__ windows, const uint32_t *__restrict__ key,
                                                                 const uint32_t *__restrict__ g, uint32_t d, int M)
{
    __shared__ __attribute__((aligned(16))) uint32_t seq[kJumpSeq];
    __shared__ uint32_t partial[kMtThreads / 64][640];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const uint32_t jump = blockIdx.x / M, slice = blockIdx.x % M;
    const uint32_t a = jump * 2u * d, dst = a + d;
    if (wave == 0) {
        // V_a: entry a of the table; V_0 = x[1 .. 624] comes from the key itself (one recurrence step for x[624])
        if (a) {
            for (int i = lane; i < kMtN; i += 64) seq[i] = windows[static_cast<uint64_t>(a) * kMtN + i];
        } else {
            for (int i = lane; i < kMtN - 1; i += 64) seq[i] = key[i + 1];
            if (lane == 0) seq[kMtN - 1] = mt_next(key[0], key[1], key[kMtM]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int b = 0; b + 1 < kJumpSeqBlocks; b++) mt_twist_wave(seq + b * kMtN, seq + (b + 1) * kMtN, lane);
    }
    __syncthreads();
    // this workgroup's coefficient words [wlo, whi) of g (bit k % 32 of word k / 32), dealt round-robin to its waves; a lane owns the
    // output words lane, lane + 64, ... (for a fixed coefficient the wave reads consecutive LDS words: no bank conflicts)
    const int wlo = kMtN * static_cast<int>(slice) / M, whi = kMtN * static_cast<int>(slice + 1) / M;
    uint32_t acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int gw = wlo + wave; gw < whi; gw += kMtThreads / 64) {
        uint32_t bits = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(g[gw])));
        while (bits) {
            const int kk = 32 * gw + __builtin_ctz(bits);
            bits &= bits - 1;
#pragma unroll
            for (int i = 0; i < 10; i++)
                if (lane + 64 * i < kMtN) acc[i] ^= seq[kk + lane + 64 * i];
        }
    }
#pragma unroll
    for (int i = 0; i < 10; i++) partial[wave][lane + 64 * i] = acc[i];
    __syncthreads();
    if (t < kMtN) {
        uint32_t x = 0;
#pragma unroll
        for (int w = 0; w < kMtThreads / 64; w++) x ^= partial[w][t];
        if (x) atomicXor(&windows[static_cast<uint64_t>(dst) * kMtN + t], x);
    }
}

// ---- host: the jump polynomials ------------------------------------------------------------------------------------------------
constexpr int kMtDeg = 19937, kJumpLevels = 12;
constexpr uint64_t kSubWords = 1ull << 17;          // S: words per substream (65,536 doubles)
// exponents of the characteristic polynomial of MT19937's state transition (found with Berlekamp-Massey on an output bit sequence,
// tools/mt_jump_poly.py; verified below, at table-build time, against the generator itself)
const uint16_t kPhiExp[135] = {
    0, 1189, 1416, 1585, 1643, 1870, 2493, 2773, 3000, 3227, 3454, 3681, 3908, 4135, 4362, 4753, 5661, 6337, 6569, 7129, 7477, 7525, 7583,
    7752, 7979, 8206, 9505, 9901, 9969, 10128, 10693, 10761, 10920, 11089, 11147, 11157, 11215, 11321, 11374, 11384, 11485, 11611, 11712,
    11717, 11838, 11881, 11944, 11997, 12277, 12335, 12393, 12504, 12509, 12620, 12673, 12731, 12736, 12789, 12905, 12958, 12963, 13137,
    13185, 13190, 13243, 13301, 13412, 13528, 13533, 13639, 13697, 13760, 13813, 13866, 14093, 14151, 14209, 14320, 14325, 14436, 14547,
    14552, 14605, 14721, 14774, 14779, 14953, 15001, 15006, 15059, 15117, 15228, 15344, 15349, 15455, 15513, 15576, 15629, 15682, 15909,
    15967, 16025, 16136, 16141, 16252, 16363, 16368, 16421, 16537, 16590, 16595, 16817, 16822, 16875, 16933, 17044, 17160, 17271, 17329,
    17445, 17498, 17725, 17783, 17841, 17952, 18068, 18179, 18237, 18406, 18633, 18691, 18860, 19087, 19314, 19937};

struct JumpTables {
    bool ok = false;
    std::vector<uint32_t> g;                 // kJumpLevels x 624 words: level j = x^(S 2^j) mod phi, bit k % 32 of word k / 32
    uint32_t *dev[64] = {};
};

inline uint64_t spread_bits(uint32_t v)
{
    uint64_t x = v;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull; x = (x | (x << 8)) & 0x00FF00FF00FF00FFull; x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull; x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}

// p <- p^2 mod phi; p: kMtDeg bits in 64-bit words
void poly_square_mod(std::vector<uint64_t> &p)
{
    const size_t W = p.size();
    std::vector<uint64_t> q(2 * W + 1, 0);
    for (size_t i = 0; i < W; i++) { q[2 * i] = spread_bits(static_cast<uint32_t>(p[i])); q[2 * i + 1] = spread_bits(static_cast<uint32_t>(p[i] >> 32)); }
    for (int dgr = 2 * (kMtDeg - 1); dgr >= kMtDeg; dgr--) {
        if (!((q[dgr >> 6] >> (dgr & 63)) & 1)) continue;
        const int sh = dgr - kMtDeg;                      // x^dgr = x^sh * x^kMtDeg = x^sh * (phi - x^kMtDeg)
        for (uint16_t e : kPhiExp) { const int bit = sh + e; q[bit >> 6] ^= 1ull << (bit & 63); }
    }
    for (size_t i = 0; i < W; i++) p[i] = q[i];
}

void host_twist(const uint32_t *o, uint32_t *nw)
{
    for (int k = 0; k < kMtN; k++) {
        const uint32_t nxt = k + 1 < kMtN ? o[k + 1] : nw[0];
        const uint32_t far = k + kMtM < kMtN ? o[k + kMtM] : nw[k + kMtM - kMtN];
        const uint32_t y = (o[k] & 0x80000000u) | (nxt & 0x7fffffffu);
        nw[k] = far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
}

JumpTables &jump_tables()
{
    static JumpTables tb;
    static std::once_flag once;
    std::call_once(once, [] {
        const size_t W = (kMtDeg + 63) / 64;
        std::vector<uint64_t> p(W, 0);
        p[0] = 2;                                                     // x
        uint64_t J = 1;
        while (J < kSubWords) { poly_square_mod(p); J *= 2; }         // x^S
        tb.g.assign(static_cast<size_t>(kJumpLevels) * kMtN, 0);
        for (int lvl = 0; lvl < kJumpLevels; lvl++) {
            for (int k = 0; k < kMtDeg; k++)
                if ((p[k >> 6] >> (k & 63)) & 1) tb.g[static_cast<size_t>(lvl) * kMtN + (k >> 5)] |= 1u << (k & 31);
            if (lvl + 1 < kJumpLevels) poly_square_mod(p);
        }
        // self check of level 0 against the generator: a test state, its stream, three words S ahead
        std::vector<uint32_t> seq(kSubWords + kMtDeg + 3 * kMtN);
        seq[0] = 19650218u;
        for (int i = 1; i < kMtN; i++) seq[i] = 1812433253u * (seq[i - 1] ^ (seq[i - 1] >> 30)) + static_cast<uint32_t>(i);
        for (size_t b = 0; (b + 2) * kMtN <= seq.size(); b++) host_twist(&seq[b * kMtN], &seq[(b + 1) * kMtN]);
        bool good = true;
        for (int w : {0, 1, 622}) {
            uint32_t acc = 0;
            for (int k = 0; k < kMtDeg; k++)
                if ((tb.g[k >> 5] >> (k & 31)) & 1) acc ^= seq[1 + k + w];
            good = good && acc == seq[1 + kSubWords + w];
        }
        tb.ok = good;
    });
    return tb;
}

}  // namespace

// Host only (no device needed): builds the jump polynomials and checks them against the generator.
extern "C" int flashe_mt19937_jump_selfcheck(void) { return jump_tables().ok ? FLASHE_OK : FLASHE_EIO; }

// One pass of flashe_mt19937_random_dev: how many of the n_left doubles it takes from stream position pos0 (0 .. 624) and into how
// many substreams it cuts them.  The jump tree has kJumpLevels levels, so a pass may span at most 2^kJumpLevels substreams WHATEVER
// pos0 is: the cap leaves room for the up to 624 words in front of the first draw (with cap = S * 2^L / 2 and pos0 >= 2 the
// last pass of a 2^28-draw call needed 2^L + 1 substreams and read one level past the table).
static void mt_plan_pass(uint64_t pos0, uint64_t n_left, bool serial_only, uint64_t *now_out, uint32_t *P_out)
{
    const uint64_t cap = ((static_cast<uint64_t>(kSubWords) << kJumpLevels) - kMtN) / 2;
    const uint64_t now = n_left < cap ? n_left : cap;
    const uint64_t end = pos0 + 2 * now;
    uint32_t P = serial_only ? 1u : static_cast<uint32_t>((end - 1 + kSubWords - 1) / kSubWords);
    if (P < 1) P = 1;
    *now_out = now;
    *P_out = P;
}

// Host only: the pass plan of a call with n draws from stream position pos -- the largest substream count of any pass and the number
// of substreams the jump tables can start (tests drive this at n around 2^28, where a pass is full).
extern "C" int flashe_mt19937_plan(uint32_t pos, uint64_t n, uint32_t *max_substreams, uint32_t *substreams_available, uint32_t *passes)
{
    if (pos > 624u || !max_substreams || !substreams_available || !passes) return FLASHE_EINVAL;
    uint64_t p = pos;
    uint32_t mx = 0, cnt = 0;
    while (n) {
        uint64_t now;
        uint32_t P;
        mt_plan_pass(p, n, false, &now, &P);
        mx = P > mx ? P : mx;
        const uint64_t end = p + 2 * now, tfin = (end - 1) / kMtN;
        p = end - tfin * kMtN;
        n -= now;
        cnt++;
    }
    *max_substreams = mx; *substreams_available = 1u << kJumpLevels; *passes = cnt;
    return FLASHE_OK;
}

extern "C" int flashe_mt19937_random_dev(flashe_ctx *ctx, uint32_t key[624], uint32_t *pos, uint64_t n, double *u_dev)
{
    CHECK_CTX(ctx);
    if (!key || !pos || *pos > 624u || (n && !u_dev)) return fail(ctx, FLASHE_EINVAL, "flashe_mt19937_random_dev: bad arguments");
    if (n == 0) return FLASHE_OK;
    if (ctx->capturing) return fail(ctx, FLASHE_EINVAL, "flashe_mt19937_random_dev is synchronous (it returns the advanced state): not capturable");
#ifdef FLASHE_TUNING
    static const bool serial_only = [] { const char *e = getenv("FLASHE_MT_PARALLEL"); return e && atoi(e) == 0; }();
#else
    constexpr bool serial_only = false;
#endif
    // scratch: [0, 625) the state in (key + pos), [640, 1264) the state out, then the windows
    uint32_t host[625];
    memcpy(host, key, 624 * sizeof(uint32_t));
    host[624] = *pos;
    while (n) {
        uint64_t now;
        uint32_t P;
        const uint64_t pos0 = host[624];
        mt_plan_pass(pos0, n, serial_only, &now, &P);
        const uint64_t end = pos0 + 2 * now;
        if (P > (1u << kJumpLevels)) return fail(ctx, FLASHE_EIO, "MT19937: a pass of %u substreams exceeds the %d-level jump table", P, kJumpLevels);
        JumpTables *jt = nullptr;
        if (P > 1) {
            jt = &jump_tables();
            if (!jt->ok) return fail(ctx, FLASHE_EIO, "MT19937 jump polynomials failed their self check");
            if (ctx->device < 0 || ctx->device >= 64) P = 1;
        }
        const size_t words = 1280 + static_cast<size_t>(P) * kMtN;
        if (ctx->mt_ws.cap < words * sizeof(uint32_t)) {             // scratch kept by the ctx: a hipMalloc + hipFree pair per call costs more than the call
            if (ctx->mt_ws.p) { HIP_TRY(ctx, hipStreamSynchronize(ctx->env.stream)); HIP_TRY(ctx, hipFree(ctx->mt_ws.p)); ctx->mt_ws.p = nullptr; ctx->mt_ws.cap = 0; }
            const size_t capb = std::max<size_t>(words * sizeof(uint32_t), 1u << 20);
            HIP_TRY(ctx, hipMalloc(&ctx->mt_ws.p, capb));
            ctx->mt_ws.cap = capb;
        }
        uint32_t *st = static_cast<uint32_t *>(ctx->mt_ws.p);
        hipError_t e = hipMemcpyAsync(st, host, sizeof host, hipMemcpyHostToDevice, ctx->env.stream);
        if (e == hipSuccess && P > 1) {
            uint32_t *&gdev = jt->dev[ctx->device];
            if (!gdev) {
                e = hipMalloc(&gdev, jt->g.size() * sizeof(uint32_t));
                if (e == hipSuccess) e = hipMemcpy(gdev, jt->g.data(), jt->g.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
                if (e != hipSuccess) gdev = nullptr;
            }
            uint32_t *windows = st + 1280;
            if (e == hipSuccess) e = hipMemsetAsync(windows, 0, static_cast<size_t>(P) * kMtN * sizeof(uint32_t), ctx->env.stream);
            int top = 0;
            while ((1u << top) < P) top++;
            for (int lvl = top - 1; lvl >= 0 && e == hipSuccess; lvl--) {
                const uint32_t d = 1u << lvl;
                if (P <= d) continue;
                const uint32_t jumps = (P - 1 - d) / (2 * d) + 1;
                int M = static_cast<int>(ctx->env.num_cus / jumps);
                M = M < 1 ? 1 : M > 16 ? 16 : M;
                hipLaunchKernelGGL(mt19937_jump_kernel, dim3(jumps * M), dim3(kMtThreads), 0, ctx->env.stream, windows, st,
                                   gdev + static_cast<size_t>(lvl) * kMtN, d, M);
                e = hipGetLastError();
            }
        }
        const uint64_t tfin = (end - 1) / kMtN;
        if (e == hipSuccess) {
            MtGen a{st, st + 1280, pos0, now, kSubWords, P, tfin, st + 640};
            hipLaunchKernelGGL(mt19937_generate_kernel, dim3(P), dim3(kMtThreads), 0, ctx->env.stream, a, u_dev);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(host, st + 640, 624 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->env.stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->env.stream);
        HIP_TRY(ctx, e);
        host[624] = static_cast<uint32_t>(end - tfin * kMtN);
        n -= now;
        u_dev += now;
    }
    memcpy(key, host, 624 * sizeof(uint32_t));
    *pos = host[624];
    return FLASHE_OK;
}
