// codec.hip — the bit codec around the comm chain on the device, frame-batched:
//   stream_encoder  lib/stream_encoder_impl.cc:76-270 + lib/utils.cc:26-290 : PDU bytes -> CRC-32 -> bits -> scrambler ->
//                   K=7 (0155, 0117) convolutional code -> puncturing -> constellation points
//   stream_decoder  lib/stream_decoder_impl.cc:205-435 + lib/viterbi_decoder.cc:62-330 : hard decisions -> depuncture ->
//                   windowed Viterbi (byte metrics, 8-bit path chunks, traceback over 5 / 10 chunks) -> descrambler -> CRC
// Integer work, bit-exact with the reference arithmetic (the __m128i byte lanes of the SSE2 decoder are one trellis state
// each, here one lane of a wave64 each).  One workgroup (encoder) / one wave (decoder) per frame: frames are the
// parallel axis, the trellis recursion inside a frame is sequential by nature.
#include "jrc_internal.h"

#define CODEC_MAX_PAYLOAD 3100                                              // lib/utils.h MAX_PAYLOAD_SIZE
#define CODEC_MAX_SYM (((16 + 8 * CODEC_MAX_PAYLOAD + 6) / 24) + 1)         // lib/utils.h MAX_SYM
#define CODEC_MAX_DBPS 4096                                                  // data bits per OFDM symbol this build sizes its LDS for
#define CODEC_MAX_DATA_BITS (16 + 8 * CODEC_MAX_PAYLOAD + 6 + CODEC_MAX_DBPS) // n_data_bits < payload bits + one symbol

struct McsParams { int n_bpsc, n_cbps, n_dbps, half_rate; };

__host__ __device__ static inline bool mcs_params(int mcs, int n_dc, McsParams& p)   // ofdm_mcs (lib/utils.cc:55-111)
{
    if (mcs < 0 || mcs > 5) return false;
    p.n_bpsc = (mcs < 2) ? 1 : ((mcs < 4) ? 2 : 4);
    p.n_cbps = n_dc * p.n_bpsc;
    p.half_rate = (mcs & 1) == 0;
    p.n_dbps = p.half_rate ? p.n_cbps / 2 : p.n_cbps * 3 / 4;
    return p.n_dbps >= 1 && p.n_dbps <= CODEC_MAX_DBPS;
}
__host__ __device__ static inline int n_sym_for(int data_size_byte, int n_dbps)      // packet_param (:30)
{
    const int bits = 16 + 8 * data_size_byte + 6;
    return (bits + n_dbps - 1) / n_dbps;               // == (int)ceil(bits / (double)n_dbps) for these magnitudes
}

__device__ __forceinline__ unsigned crc_table_entry(unsigned i)
{
    unsigned c = i;
#pragma unroll
    for (int b = 0; b < 8; b++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
    return c;
}

__device__ __forceinline__ float2 constellation_point(int bpsc, int v)     // gr::digital bpsk / qpsk (/2, :218-221) / 16qam
{
    if (bpsc == 1) return make_float2(v ? 1.0f : -1.0f, 0.0f);
    if (bpsc == 2) {
        const float s = 0.707107f;
        return make_float2(((v & 1) ? s : -s) / 2.0f, ((v & 2) ? s : -s) / 2.0f);
    }
    const float level = 0.316227766016837933f;                             // sqrt(float(0.1)) rounded to float
    const float a = (v & 2) ? 1.0f : 3.0f, b = (v & 8) ? 1.0f : 3.0f;
    return make_float2(((v & 1) ? a : -a) * level, ((v & 4) ? b : -b) * level);
}
__device__ __forceinline__ int constellation_decide(int bpsc, float2 z)
{
    if (bpsc == 1) return z.x > 0;
    if (bpsc == 2) return 2 * (z.y > 0) + (z.x > 0);
    const float level = 0.316227766016837933f;
    return (z.x > 0) | ((fabsf(z.x) < 2 * level) << 1) | ((z.y > 0) << 2) | ((fabsf(z.y) < 2 * level) << 3);
}

// ---- encoder: one workgroup per PDU ---------------------------------------------------------------
__global__ __launch_bounds__(256) void stream_encode_kernel(int mcs, int n_dc, const unsigned char* __restrict__ psdu, long psdu_stride,
                                                            const int* __restrict__ lens, const unsigned char* __restrict__ scr_init,
                                                            float2* __restrict__ out, long sym_stride, int* __restrict__ n_sym_out)
{
    __shared__ unsigned crc_tab[256];
    __shared__ unsigned char seq[128];
    __shared__ unsigned fcs_s;
    extern __shared__ unsigned char sbits[];                               // scrambled data bits, one per byte
    const int f = blockIdx.x, tid = threadIdx.x;
    const int len = lens[f];
    McsParams p;
    mcs_params(mcs, n_dc, p);
    if (len < 0 || len + 4 > CODEC_MAX_PAYLOAD) {                          // :139-143: "Data Packet too Large" -> nothing is produced
        if (tid == 0) n_sym_out[f] = 0;
        return;
    }
    const unsigned char* pk = psdu + (size_t)f * psdu_stride;
    const int dsb = len + 4;
    const int n_sym = n_sym_for(dsb, p.n_dbps);
    const int ndb = n_sym * p.n_dbps, npad = ndb - (16 + 8 * dsb + 6);
    crc_tab[tid] = crc_table_entry(tid);
    __syncthreads();
    if (tid == 0) {                                                        // boost::crc_32_type over the PDU (:150-153)
        unsigned c = 0xFFFFFFFFu;
        for (int i = 0; i < len; i++) c = crc_tab[(c ^ pk[i]) & 0xff] ^ (c >> 8);
        fcs_s = c ^ 0xFFFFFFFFu;
    }
    if (tid == 64) {                                                       // scrambler sequence (period 127) for this PDU's state (:151-162)
        int state = (signed char)scr_init[f];
        for (int i = 0; i < 127; i++) {
            const int fb = (!!(state & 64)) ^ (!!(state & 8));
            seq[i] = (unsigned char)fb;
            state = ((state << 1) & 0x7e) | fb;
        }
    }
    __syncthreads();
    const unsigned fcs = fcs_s;
    for (int i = tid; i < ndb; i += 256) {                                 // generate_bits + scramble + reset_tail_bits
        int bit = 0;
        if (i >= 16 && i < 16 + 8 * dsb) {
            const int by = (i - 16) >> 3, b = (i - 16) & 7;
            const unsigned v = by < len ? pk[by] : (fcs >> (8 * (by - len))) & 0xff;
            bit = (v >> b) & 1;
        }
        bit ^= seq[i % 127];
        const int tail = ndb - npad - 6;
        if (i >= tail && i < tail + 6) bit = 0;
        sbits[i] = (unsigned char)bit;
    }
    __syncthreads();
    float2* o = out + (size_t)f * sym_stride;
    const int nsymb = n_sym * n_dc;
    for (int j = tid; j < nsymb; j += 256) {                               // convolutional code + puncturing + split + mapping
        int v = 0;
        for (int k = 0; k < p.n_bpsc; k++) {
            const int c = j * p.n_bpsc + k;                                // index in the punctured stream
            int e = c;
            if (!p.half_rate) { const int q = c >> 2, r = c & 3; e = 6 * q + (r == 3 ? 5 : r); }   // kept positions 0,1,2,5 of every 6 (:211-216)
            const int i = e >> 1;
            int reg = 0;                                                   // encoder register after bit i: bit d = in[i - d]
#pragma unroll
            for (int d = 0; d < 7; d++) if (i - d >= 0) reg |= sbits[i - d] << d;
            const int taps = (e & 1) ? 0117 : 0155;
            v |= (__popc(reg & taps) & 1) << k;
        }
        o[j] = constellation_point(p.n_bpsc, v);
    }
    if (tid == 0) n_sym_out[f] = nsymb;
}

// ---- decoder: one wave per frame, lane = trellis state -----------------------------------------------
__device__ __forceinline__ int wave_max_i32(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ int wave_min_i32(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off));
    return v;
}

#define DEC_WAVES 4      // frames per workgroup

// Hard decisions never touch memory: every 64 coded bits, lane i loads the symbol that carries coded bit c0 + i (issued one
// block ahead), decides it (decision_maker, :166-169) and a wave ballot turns the 64 decisions into one scalar bit word.
// Decoded bytes are descrambled and fed to the CRC as they leave the traceback, so the only LDS is the path ring.
__global__ __launch_bounds__(64 * DEC_WAVES) void stream_decode_kernel(int n_dc, int n_frames, const float2* __restrict__ sym, long sym_stride,
                                                                       const int* __restrict__ mcs_arr, const int* __restrict__ bytes_arr,
                                                                       unsigned char* __restrict__ payload, long payload_stride,
                                                                       int* __restrict__ status)
{
    __shared__ unsigned char ring[DEC_WAVES][10][64];
    __shared__ unsigned crc_tab[256];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // the wave index is uniform, but only this tells the compiler: without it
                                                                              // every per-frame quantity below is compiled as exec-masked vector code
    const int f = blockIdx.x * DEC_WAVES + w;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) crc_tab[i] = crc_table_entry(i);
    for (int i = lane; i < 10 * 64; i += 64) (&ring[w][0][0])[i] = 0;                   // d_ppresult zeroed (:333-337)
    __syncthreads();
    if (f >= n_frames) return;
    // everything that depends only on the frame is wave-uniform: keep it in scalar registers so the control flow below is
    // scalar branches, not exec-masked vector code
    const int mcs = __builtin_amdgcn_readfirstlane(mcs_arr[f]), dsb = __builtin_amdgcn_readfirstlane(bytes_arr[f]);
    McsParams p;
    if (!mcs_params(mcs, n_dc, p) || dsb < 0 || dsb > CODEC_MAX_PAYLOAD || n_sym_for(dsb, p.n_dbps) > CODEC_MAX_SYM) {   // :133-146
        if (lane == 0) status[f] = -1;
        return;
    }
    const int n_sym = n_sym_for(dsb, p.n_dbps);
    const int ndb = n_sym * p.n_dbps, n_coded = n_sym * p.n_cbps;
    const float2* s = sym + (size_t)f * sym_stride;
    unsigned char* pl = payload + (size_t)f * payload_stride;
    const int bpsc = p.n_bpsc, half_rate = p.half_rate;

    // coded-bit source
    auto load_sym = [&](int c0) -> float2 {     // unconditional (index clamped): a select on the loaded value would force the wait right here,
        const int c = max(0, min(c0 + lane, n_coded - 1));     // and the load is issued a block ahead precisely to avoid that; decide_word
        return s[c / bpsc];                                    // zeroes the bits past the frame
    };
    auto decide_word = [&](float2 z, int c0) -> unsigned long long {
        const int c = c0 + lane;
        const int bit = (c < n_coded) ? ((constellation_decide(bpsc, z) >> (c % bpsc)) & 1) : 0;
        return __ballot(bit);
    };
    float2 z_pending = load_sym(0);
    unsigned long long word = decide_word(z_pending, 0);
    z_pending = load_sym(64);
    int wbase = 0, wpos = 0;                  // wpos = bits of `word` already consumed
    auto next_bit = [&]() -> int {
        const int b = (int)((word >> wpos) & 1ull);
        wpos++;
        return b;
    };

    // trellis: lane = new state; butterfly k = lane >> 1 reads old states k and k + 32 (viterbi_butterfly2_sse2, :87-180)
    const int k = lane >> 1, odd = lane & 1;
    const int bt0 = __popc((2 * k) & 0x6d) & 1, bt1 = __popc((2 * k) & 0x4f) & 1;       // d_branchtab27_sse2 (:323-326)
    const int nt = half_rate ? 5 : 10;                                                  // reset() (:293-315)
    unsigned st = 0;                                                                    // metric | path << 8
    const int n_calls = nt + (ndb + 7) / 8;                                             // get_output calls until n_decoded >= n_data_bits
    const int n_steps = 6 + 8 * (n_calls - 1);
    int store_pos = 0, out_count = 0, phase3 = 0;
    int lfsr = 0;
    unsigned crc = 0xFFFFFFFFu;
    for (int t = 0; t < n_steps; t++) {
        int s0 = 0, s1 = 0;                                                             // past the end of the frame: 0 (fresh buffers)
        if (t < ndb) {
            // a word runs out only on a step boundary: 64 bits = 32 steps at rate 1/2, 48 steps (phase 0) at rate 3/4
            if (wpos == 64) {
                wbase += 64; wpos = 0;
                word = decide_word(z_pending, wbase);
                z_pending = load_sym(wbase + 64);
            }
            if (half_rate) { s0 = next_bit(); s1 = next_bit(); }
            else {                                                                      // depuncture (:236-252): pattern 1,1,1,0,0,1
                if (phase3 == 0) { s0 = next_bit(); s1 = next_bit(); }
                else if (phase3 == 1) { s0 = next_bit(); s1 = 2; }
                else { s0 = 2; s1 = next_bit(); }
                phase3 = (phase3 == 2) ? 0 : phase3 + 1;
            }
        }
        int metsvm, metsv;
        if (s0 == 2) { metsvm = bt1 ^ s1; metsv = 1 - metsvm; }
        else if (s1 == 2) { metsvm = bt0 ^ s0; metsv = 1 - metsvm; }
        else { metsvm = (bt0 ^ s0) + (bt1 ^ s1); metsv = 2 - metsvm; }
        const unsigned a = __shfl(st, k), b = __shfl(st, k + 32);
        const int ma = a & 0xff, mb = b & 0xff;
        // even new state: m0 = ma + metsv vs m1 = mb + metsvm ; odd: m2 = ma + metsvm vs m3 = mb + metsv
        const int x = (ma + (odd ? metsvm : metsv)) & 0xff, y = (mb + (odd ? metsv : metsvm)) & 0xff;
        const int dec = (signed char)((x - y) & 0xff) > 0;                              // _mm_cmpgt_epi8(_mm_sub_epi8(m0, m1), 0)
        const unsigned pa = ((a >> 8) << 1) & 0xff, pb = ((((b >> 8) << 1) & 0xff) + 1) & 0xff;
        st = (unsigned)(dec ? x : y) | ((dec ? pa : pb) << 8);
        if (t >= 5 && ((t - 5) & 7) == 0) {                                             // viterbi_get_output_sse2 (:183-225) after steps 6, 14, 22, ...
            store_pos = (store_pos + 1 == nt) ? 0 : store_pos + 1;
            const int metric = st & 0xff;
            ring[w][store_pos][lane] = (unsigned char)(st >> 8);
            const int best = wave_max_i32(metric), mn = wave_min_i32(metric);
            int beststate = __ffsll((unsigned long long)__ballot(metric == best)) - 1;  // first maximum (strict > in the scan)
            __builtin_amdgcn_wave_barrier();
            int pos = store_pos;
            for (int i = 0; i < nt - 1; i++) {
                beststate = ring[w][pos][beststate] >> 2;
                pos = (pos == 0) ? nt - 1 : pos - 1;
            }
            const int c = __builtin_amdgcn_readfirstlane((int)ring[w][pos][beststate]);
            st = (unsigned)((metric - mn) & 0xff);                                      // paths zeroed, metrics renormalised
            if (out_count >= nt) {                                                      // decoded bits, MSB first (:277-281)
                const int j = out_count - nt;                                           // byte j = stream bits 8j .. 8j+7
                int ob = 0;
                if (j == 0) {                                                           // descramble (:406-433): state from bits 0..6
                    lfsr = (c >> 1) & 0x7f;
                    const int fb = (!!(lfsr & 64)) ^ (!!(lfsr & 8));                    // bit 7 belongs to out_bytes[0] (unused)
                    lfsr = ((lfsr << 1) & 0x7e) | fb;
                } else {
#pragma unroll
                    for (int bb = 0; bb < 8; bb++) {
                        const int fb = (!!(lfsr & 64)) ^ (!!(lfsr & 8));
                        ob |= (fb ^ ((c >> (7 - bb)) & 1)) << bb;
                        lfsr = ((lfsr << 1) & 0x7e) | fb;
                    }
                }
                const int by = j - 2;                                                   // out_bytes + 2 = PSDU incl. CRC
                if (by >= 0 && by < dsb) {
                    crc = crc_tab[(crc ^ (unsigned)ob) & 0xff] ^ (crc >> 8);
                    if (by < dsb - 4 && lane == 0) pl[by] = (unsigned char)ob;
                }
            }
            out_count++;
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (lane == 0) status[f] = ((crc ^ 0xFFFFFFFFu) == 558161692u) ? 1 : 0;              // :245-246
}

// ---- decoder, two frames per wave -----------------------------------------------------------------------------------
// The add-compare-select of a trellis step is a dozen 8-bit operations per state; a 32-bit lane register holds the state of TWO
// frames (bits 0-7 / 16-23 path metrics, bits 8-15 / 24-31 path chunks), so the two cross-lane fetches, the branch-metric
// arithmetic and the select of a step serve both frames: the halves never carry into each other (metric + increment < 2^9 is
// masked back to 8 bits, the compare subtracts with a guard bit per half).  Everything per frame that is wave-uniform — the bit
// FIFO, depuncturing phase, traceback position, descrambler, CRC — stays on the scalar unit, once per frame.  Same arithmetic,
// same outputs as stream_decode_kernel (which keeps the single-frame host call).
#define DEC2_WAVES 4      // waves per workgroup, two frames each

typedef unsigned short dec_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(dec_us2, a), __builtin_bit_cast(dec_us2, b)));
}
__device__ __forceinline__ unsigned pk_min_u16(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(dec_us2, a), __builtin_bit_cast(dec_us2, b)));
}

// wave-wide maximum / minimum of both 16-bit halves with DPP row shifts and row broadcasts (no LDS crossbar): the result sits in lane 63
template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned dpp_pk_max(unsigned v) { return pk_max_u16(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROWMASK, 0xf, false)); }
template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned dpp_pk_min(unsigned v) { return pk_min_u16(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROWMASK, 0xf, false)); }
__device__ __forceinline__ unsigned wave_pk_max_u16(unsigned v)
{
    v = dpp_pk_max<0x111, 0xf>(v); v = dpp_pk_max<0x112, 0xf>(v); v = dpp_pk_max<0x114, 0xf>(v); v = dpp_pk_max<0x118, 0xf>(v);   // row_shr 1, 2, 4, 8
    v = dpp_pk_max<0x142, 0xa>(v); v = dpp_pk_max<0x143, 0xc>(v);                                                                 // row_bcast 15, 31
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_pk_min_u16(unsigned v)
{
    v = dpp_pk_min<0x111, 0xf>(v); v = dpp_pk_min<0x112, 0xf>(v); v = dpp_pk_min<0x114, 0xf>(v); v = dpp_pk_min<0x118, 0xf>(v);
    v = dpp_pk_min<0x142, 0xa>(v); v = dpp_pk_min<0x143, 0xc>(v);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

template <int NQ /* frames per wave: 2 = throughput (both halves of the lane register), 1 = shortest dependent chain per step */>
__global__ __launch_bounds__(64 * DEC2_WAVES) void stream_decode2_kernel(int n_dc, int n_frames, const float2* __restrict__ sym, long sym_stride,
                                                                         const int* __restrict__ mcs_arr, const int* __restrict__ bytes_arr,
                                                                         unsigned char* __restrict__ payload, long payload_stride,
                                                                         int* __restrict__ status)
{
    // decoded bytes collect in LDS and go out in one coalesced copy per frame: a byte store per output inside the loop sits in the same
    // in-order memory counter as the symbol prefetch, whose wait would then also wait for every store behind it
    __shared__ unsigned char obuf[DEC2_WAVES][2][(CODEC_MAX_PAYLOAD + 7) & ~7];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // the wave index is uniform, but only this tells the compiler: without it
                                                                              // every per-frame quantity below is compiled as exec-masked vector code
    const int f0 = (blockIdx.x * DEC2_WAVES + w) * NQ;
    if (f0 >= n_frames) return;
    // CRC-32 table in registers: entry e in lane e % 64 of tabv[e / 64], fetched with v_readlane (the index is wave-uniform)
    unsigned tabv[4];
#pragma unroll
    for (int j = 0; j < 4; j++) tabv[j] = crc_table_entry(64 * j + lane);
    auto crc_entry = [&](unsigned e) -> unsigned {
        const int l = (int)(e & 63u);
        const unsigned x0 = (unsigned)__builtin_amdgcn_readlane((int)tabv[0], l), x1 = (unsigned)__builtin_amdgcn_readlane((int)tabv[1], l);
        const unsigned x2 = (unsigned)__builtin_amdgcn_readlane((int)tabv[2], l), x3 = (unsigned)__builtin_amdgcn_readlane((int)tabv[3], l);
        const unsigned hi = e >> 6;
        return hi == 0 ? x0 : (hi == 1 ? x1 : (hi == 2 ? x2 : x3));
    };
    // descrambler (:406-433) a byte at a time: for each of the 128 register states, the eight feedback bits it emits (bit b = b-th) and
    // the state after them — entry s in lane s % 64 of scrv[s / 64]
    unsigned scrv[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        int l = 64 * j + lane, fb8 = 0;
#pragma unroll
        for (int bb = 0; bb < 8; bb++) {
            const int fb = (!!(l & 64)) ^ (!!(l & 8));
            fb8 |= fb << bb;
            l = ((l << 1) & 0x7e) | fb;
        }
        scrv[j] = (unsigned)fb8 | ((unsigned)l << 8);
    }
    // the last ten path chunks of every state (both frames per register), newest first: d_ppresult (:333-337) as a shift register, so
    // the traceback walks registers with v_readlane instead of chasing bytes through LDS
    unsigned rr[10];
#pragma unroll
    for (int i = 0; i < 10; i++) rr[i] = 0;

    // per-frame, wave-uniform
    int act[2], bpsc[2], half_rate[2], nt[2], ndb[2], n_coded[2], dsb[2], n_steps[2];
    const float2* sp[2];
    unsigned char* pl[2];
    int steps = 0;
    act[1] = 0; bpsc[1] = 1; half_rate[1] = 1; nt[1] = 5; ndb[1] = 0; n_coded[1] = 0; dsb[1] = 0; n_steps[1] = 0; sp[1] = sym; pl[1] = payload;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const int f = f0 + q;
        act[q] = f < n_frames;
        bpsc[q] = 1; half_rate[q] = 1; nt[q] = 5; ndb[q] = 0; n_coded[q] = 0; dsb[q] = 0; n_steps[q] = 0; sp[q] = sym; pl[q] = payload;
        if (act[q]) {
            const int mcs = __builtin_amdgcn_readfirstlane(mcs_arr[f]);
            dsb[q] = __builtin_amdgcn_readfirstlane(bytes_arr[f]);
            McsParams p;
            if (!mcs_params(mcs, n_dc, p) || dsb[q] < 0 || dsb[q] > CODEC_MAX_PAYLOAD || n_sym_for(dsb[q], p.n_dbps) > CODEC_MAX_SYM) {   // :133-146
                if (lane == 0) status[f] = -1;
                act[q] = 0;
            } else {
                const int n_sym = n_sym_for(dsb[q], p.n_dbps);
                ndb[q] = n_sym * p.n_dbps; n_coded[q] = n_sym * p.n_cbps;
                bpsc[q] = p.n_bpsc; half_rate[q] = p.half_rate;
                nt[q] = p.half_rate ? 5 : 10;                                           // reset() (:293-315)
                const int n_calls = nt[q] + (ndb[q] + 7) / 8;                           // get_output calls until n_decoded >= n_data_bits
                n_steps[q] = 6 + 8 * (n_calls - 1);
                sp[q] = sym + (size_t)f * sym_stride;
                pl[q] = payload + (size_t)f * payload_stride;
                steps = max(steps, n_steps[q]);
            }
        }
    }
    if (!act[0] && (NQ == 1 || !act[1])) return;

    auto load_sym = [&](int q, int c0) -> float2 {  // unconditional (index clamped): a select on the loaded value would force the wait right
        const int c = max(0, min(c0 + lane, n_coded[q] - 1));  // here, and the load is issued a block ahead precisely to avoid that;
        return sp[q][c / bpsc[q]];                             // decide_word zeroes the bits past the frame
    };
    auto decide_word = [&](int q, float2 z, int c0) -> unsigned long long {
        const int c = c0 + lane;
        const int bit = (c < n_coded[q]) ? ((constellation_decide(bpsc[q], z) >> (c % bpsc[q])) & 1) : 0;
        return __ballot(bit);
    };
    // The soft-bit pair of a trellis step (depuncturing included, :236-252) is prepared a block at a time by the lanes: one 64-bit word
    // of hard decisions feeds 32 steps at rate 1/2 or 48 steps at rate 3/4 (pattern 1,1,1,0,0,1 — a block starts at phase 0); lane i
    // works out the pair of step blk_t0 + i as four flags (bit 0 / 1: the two bits, bit 2 / 3: transmitted or punctured).  A step then
    // costs the scalar unit one v_readlane per frame instead of a bit FIFO with its shifts and branches.
    float2 zp[2];
    int wbase[2] = {0, 0}, blk_t0[2] = {0, 0}, out_count[2] = {0, 0}, lfsr[2] = {0, 0};
    unsigned ctl[2];
    unsigned crc[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
    auto make_ctl = [&](int q, unsigned long long word, int t0) -> unsigned {
        int p0, p1, e0 = 1, e1 = 1;
        if (half_rate[q]) { p0 = 2 * lane; p1 = 2 * lane + 1; }
        else {
            const int g = lane / 3, ph = lane - 3 * g;
            p0 = 4 * g + (ph == 1 ? 2 : 0); p1 = 4 * g + (ph == 2 ? 3 : 1);
            e0 = ph != 2; e1 = ph != 1;
        }
        unsigned b0 = (unsigned)((word >> (p0 & 63)) & 1ull) & (unsigned)e0, b1 = (unsigned)((word >> (p1 & 63)) & 1ull) & (unsigned)e1;
        if (t0 + lane >= ndb[q]) { b0 = 0; b1 = 0; e0 = 1; e1 = 1; }                   // past the end of the frame: 0, 0 (fresh buffers)
        return b0 | (b1 << 1) | ((unsigned)e0 << 2) | ((unsigned)e1 << 3);
    };
    // two blocks of soft-bit pairs are kept (current, next) so that a group of eight steps never has to stop for a refill in the middle
    int blk_len[2];
    unsigned ctl_nxt[2];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        blk_len[q] = half_rate[q] ? 32 : 48;
        zp[q] = load_sym(q, 0);
        ctl[q] = make_ctl(q, decide_word(q, zp[q], 0), 0);
        zp[q] = load_sym(q, 64);
        ctl_nxt[q] = make_ctl(q, decide_word(q, zp[q], 64), blk_len[q]);
        zp[q] = load_sym(q, 128);
    }

    // trellis: lane = new state; butterfly k = lane >> 1 reads old states k and k + 32 (viterbi_butterfly2_sse2, :87-180)
    const int k = lane >> 1, odd = lane & 1;
    const unsigned bt0p = (unsigned)(__popc((2 * k) & 0x6d) & 1) * 0x00010001u;         // d_branchtab27_sse2 (:323-326), both halves
    const unsigned bt1p = (unsigned)(__popc((2 * k) & 0x4f) & 1) * 0x00010001u;
    unsigned st = 0;                                                                    // per half: metric | path << 8
    auto acs_step = [&](int t) {
        unsigned cc = 0;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int idx = t - blk_t0[q];
            const int v0 = __builtin_amdgcn_readlane((int)ctl[q], idx < blk_len[q] ? idx : 0);
            const int v1 = __builtin_amdgcn_readlane((int)ctl_nxt[q], idx < blk_len[q] ? 0 : idx - blk_len[q]);
            cc |= (unsigned)(idx < blk_len[q] ? v0 : v1) << (16 * q);
        }
        const unsigned s0p = cc & 0x00010001u, s1p = (cc >> 1) & 0x00010001u, e0p = (cc >> 2) & 0x00010001u, e1p = (cc >> 3) & 0x00010001u;
        // branch metrics of both frames: metsvm = (bt0 ^ s0) + (bt1 ^ s1) over the transmitted bits, metsv = (their number) - metsvm
        const unsigned m = ((bt0p ^ s0p) & e0p) + ((bt1p ^ s1p) & e1p);
        const unsigned totp = e0p + e1p;
        const unsigned xadd = odd ? m : totp - m;                                       // even new state: ma + metsv vs mb + metsvm; odd: the other way
        const unsigned yadd = totp - xadd;
        const unsigned a = __shfl(st, k), b = __shfl(st, k + 32);
        if constexpr (NQ == 1) {
            // one frame: the reference's byte compare directly — sign of the 8-bit difference — a chain of six operations behind the fetch
            const unsigned x = (a + xadd) & 0xffu, y = (b + yadd) & 0xffu;
            const bool dec = (int)__builtin_amdgcn_sbfe((int)(x - y), 0, 8) > 0;        // _mm_cmpgt_epi8(_mm_sub_epi8(m0, m1), 0)
            const unsigned pa = (a & 0x7f00u) << 1, pb = ((b & 0x7f00u) << 1) | 0x0100u;
            st = dec ? (x | pa) : (y | pb);
        } else {
            const unsigned ca = ((a + xadd) & 0x00ff00ffu) | ((a & 0x7f007f00u) << 1);
            const unsigned cb = ((b + yadd) & 0x00ff00ffu) | ((b & 0x7f007f00u) << 1) | 0x01000100u;
            // _mm_cmpgt_epi8(_mm_sub_epi8(m0, m1), 0) per half: d = (x - y) mod 256 in 1..127
            const unsigned d = (((ca & 0x00ff00ffu) | 0x01000100u) - (cb & 0x00ff00ffu)) & 0x00ff00ffu;
            const unsigned dec7 = ((d & 0x007f007fu) + 0x007f007fu) & ~d & 0x00800080u;
            const unsigned mask = (dec7 >> 7) * 0xffffu;
            st = (ca & mask) | (cb & ~mask);
        }
    };
    auto output_event = [&](int t) {                                                    // viterbi_get_output_sse2 (:183-225) after steps 6, 14, 22, ...
        {
            int live[2] = {0, 0};
#pragma unroll
            for (int q = 0; q < NQ; q++) live[q] = act[q] && t < n_steps[q];
#pragma unroll
            for (int i = 9; i > 0; i--) rr[i] = rr[i - 1];
            rr[0] = (st >> 8) & 0x00ff00ffu;
            const unsigned mp = st & 0x00ff00ffu;
            const unsigned mx = wave_pk_max_u16(mp), mn = wave_pk_min_u16(mp);
            int c[2] = {0, 0};
#pragma unroll
            for (int q = 0; q < NQ; q++)
                if (live[q]) {
                    const unsigned metric = (mp >> (16 * q)) & 0xffu, best = (mx >> (16 * q)) & 0xffu;
                    int bs = __ffsll((unsigned long long)__ballot(metric == best)) - 1;        // first maximum (strict > in the scan)
#pragma unroll
                    for (int i = 0; i < 9; i++)
                        if (i < nt[q] - 1) bs = (int)((((unsigned)__builtin_amdgcn_readlane((int)rr[i], bs) >> (16 * q)) & 0xffu) >> 2);
                    const unsigned last = half_rate[q] ? (unsigned)__builtin_amdgcn_readlane((int)rr[4], bs) : (unsigned)__builtin_amdgcn_readlane((int)rr[9], bs);
                    c[q] = (int)((last >> (16 * q)) & 0xffu);
                }
            // paths zeroed, metrics renormalised (each half >= its minimum: no borrow)
            {
                const unsigned keep = (live[0] ? 0x0000ffffu : 0u) | (live[1] ? 0xffff0000u : 0u);
                st = (st & ~keep) | ((mp - mn) & 0x00ff00ffu & keep);
            }
#pragma unroll
            for (int q = 0; q < NQ; q++)
                if (live[q]) {
                    if (out_count[q] >= nt[q]) {                                        // decoded bits, MSB first (:277-281)
                        const int j = out_count[q] - nt[q];                             // byte j = stream bits 8j .. 8j+7
                        int ob = 0;
                        if (j == 0) {                                                   // descramble (:406-433): state from bits 0..6
                            lfsr[q] = (c[q] >> 1) & 0x7f;
                            const int fb = (!!(lfsr[q] & 64)) ^ (!!(lfsr[q] & 8));      // bit 7 belongs to out_bytes[0] (unused)
                            lfsr[q] = ((lfsr[q] << 1) & 0x7e) | fb;
                        } else {
                            const unsigned e0 = (unsigned)__builtin_amdgcn_readlane((int)scrv[0], lfsr[q] & 63);
                            const unsigned e1 = (unsigned)__builtin_amdgcn_readlane((int)scrv[1], lfsr[q] & 63);
                            const unsigned e = (lfsr[q] & 64) ? e1 : e0;
                            ob = (int)((e & 0xffu) ^ (__brev((unsigned)c[q]) >> 24));      // bit b of ob = b-th feedback bit ^ bit (7 - b) of c
                            lfsr[q] = (int)(e >> 8);
                        }
                        const int by = j - 2;                                           // out_bytes + 2 = PSDU incl. CRC
                        if (by >= 0 && by < dsb[q]) {
                            crc[q] = crc_entry((crc[q] ^ (unsigned)ob) & 0xffu) ^ (crc[q] >> 8);
                            if (by < dsb[q] - 4 && lane == 0) obuf[w][q][by] = (unsigned char)ob;
                        }
                    }
                    out_count[q]++;
                }
        }
    };
    // the trellis runs in groups that end with an output: steps 0..5, then eight at a time; soft-bit blocks rotate between groups only
    auto rotate = [&](int g0) {
#pragma unroll
        for (int q = 0; q < NQ; q++)
            if (g0 >= blk_t0[q] + blk_len[q]) {
                blk_t0[q] += blk_len[q]; wbase[q] += 64;
                ctl[q] = ctl_nxt[q];
                ctl_nxt[q] = make_ctl(q, decide_word(q, zp[q], wbase[q] + 64), blk_t0[q] + blk_len[q]);
                zp[q] = load_sym(q, wbase[q] + 128);
            }
    };
#pragma unroll
    for (int i = 0; i < 6; i++) acs_step(i);
    output_event(5);
    for (int g0 = 6; g0 < steps; g0 += 8) {
        rotate(g0);
#pragma unroll
        for (int i = 0; i < 8; i++) acs_step(g0 + i);
        output_event(g0 + 7);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < NQ; q++)
        if (act[q])
            for (int i = lane; i < dsb[q] - 4; i += 64) pl[q][i] = obuf[w][q][i];
#pragma unroll
    for (int q = 0; q < NQ; q++)
        if (act[q] && lane == 0) status[f0 + q] = ((crc[q] ^ 0xFFFFFFFFu) == 558161692u) ? 1 : 0;   // :245-246
}

// ---- C ABI ------------------------------------------------------------------------------------------
extern "C" int jrc_stream_n_ofdm_sym(int mcs, int n_data_carriers, int data_size_byte)
{
    McsParams p;
    if (!mcs_params(mcs, n_data_carriers, p) || data_size_byte < 0) return JRC_ERR_INVALID_ARG;
    return n_sym_for(data_size_byte, p.n_dbps);
}

extern "C" int jrc_stream_encode_dev(jrc_ctx* ctx, int mcs, int n_data_carriers, int n_frames, const uint8_t* d_psdu, long psdu_stride,
                                     const int* d_len, const uint8_t* d_scrambler, jrc_cf32* d_out, long sym_stride, int* d_n_sym,
                                     void* stream)
{
    JRC_TRACE("jrc_stream_encode_dev");
    if (!ctx) return JRC_ERR_INVALID_ARG;
    McsParams p;
    if (!mcs_params(mcs, n_data_carriers, p) || n_data_carriers < 1)
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "stream_encoder: wrong encoding %d", mcs);       // std::invalid_argument("wrong encoding")
    if (n_frames < 0 || (n_frames > 0 && (!d_psdu || !d_len || !d_scrambler || !d_out || !d_n_sym)))
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "stream_encoder: invalid arguments");
    if (n_frames == 0) return JRC_OK;
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    JRC_BIND(ctx);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const size_t lds = (size_t)CODEC_MAX_DATA_BITS;
    hipLaunchKernelGGL(stream_encode_kernel, dim3(n_frames), dim3(256), lds, s, mcs, n_data_carriers, d_psdu, psdu_stride, d_len,
                       d_scrambler, (float2*)d_out, sym_stride, d_n_sym);
    JRC_HIP(ctx, hipGetLastError());
    return JRC_OK;
}

extern "C" int jrc_stream_decode_dev(jrc_ctx* ctx, int n_data_carriers, int n_frames, const jrc_cf32* d_sym, long sym_stride,
                                     const int* d_mcs, const int* d_data_bytes, uint8_t* d_payload, long payload_stride, int* d_status,
                                     void* stream)
{
    JRC_TRACE("jrc_stream_decode_dev");
    if (!ctx) return JRC_ERR_INVALID_ARG;
    if (n_data_carriers < 1 || n_frames < 0 || (n_frames > 0 && (!d_sym || !d_mcs || !d_data_bytes || !d_payload || !d_status)))
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "stream_decoder: invalid arguments");
    if (n_frames == 0) return JRC_OK;
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    JRC_BIND(ctx);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    // few frames: one per wave (the decoder is a latency chain; twice the waves, each with the shorter chain); many: two per wave
    int nq = ctx->tune.dec_frames_per_wave;
    if (nq != 1 && nq != 2) nq = n_frames >= 4 * ctx->n_cus * 4 * 2 ? 2 : 1;
    if (nq == 2) {
        const int per_wg = 2 * DEC2_WAVES;
        hipLaunchKernelGGL(stream_decode2_kernel<2>, dim3((n_frames + per_wg - 1) / per_wg), dim3(64 * DEC2_WAVES), 0, s, n_data_carriers,
                           n_frames, (const float2*)d_sym, sym_stride, d_mcs, d_data_bytes, d_payload, payload_stride, d_status);
    } else if (!ctx->tune.dec_single) {
        hipLaunchKernelGGL(stream_decode2_kernel<1>, dim3((n_frames + DEC2_WAVES - 1) / DEC2_WAVES), dim3(64 * DEC2_WAVES), 0, s, n_data_carriers,
                           n_frames, (const float2*)d_sym, sym_stride, d_mcs, d_data_bytes, d_payload, payload_stride, d_status);
    } else {
        hipLaunchKernelGGL(stream_decode_kernel, dim3((n_frames + DEC_WAVES - 1) / DEC_WAVES), dim3(64 * DEC_WAVES), 0, s, n_data_carriers,
                           n_frames, (const float2*)d_sym, sym_stride, d_mcs, d_data_bytes, d_payload, payload_stride, d_status);
    }
    JRC_HIP(ctx, hipGetLastError());
    return JRC_OK;
}

// host-buffer forms (one PDU / one frame), as the blocks' general_work would call them
extern "C" int jrc_stream_encode(jrc_ctx* ctx, int mcs, int n_data_carriers, const uint8_t* psdu, int len, int scrambler_init,
                                 jrc_cf32* out_symbols, int out_capacity)
{
    if (!ctx || len < 0 || (len > 0 && !psdu)) return JRC_ERR_INVALID_ARG;
    McsParams p;
    if (!mcs_params(mcs, n_data_carriers, p)) return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "stream_encoder: wrong encoding %d", mcs);
    if (len + 4 > CODEC_MAX_PAYLOAD) return 0;                                              // :139-143
    const int nsymb = n_sym_for(len + 4, p.n_dbps) * n_data_carriers;
    if (!out_symbols || out_capacity < nsymb) return jrc_fail(ctx, JRC_ERR_SHORT_INPUT, "stream_encoder: output holds %d symbols, %d needed", out_capacity, nsymb);
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t in_bytes = ((size_t)len + 15) & ~(size_t)15, meta = 16, out_bytes = sizeof(float2) * (size_t)nsymb;
    JRC_TRY(jrc_ensure_pinned(ctx, in_bytes + meta + out_bytes + 16));
    JRC_TRY(jrc_ensure_scratch(ctx, 0, in_bytes + meta));
    JRC_TRY(jrc_ensure_scratch(ctx, 1, out_bytes + 16));
    char* pin = (char*)ctx->pinned;
    if (len) memcpy(pin, psdu, (size_t)len);
    int* meta_h = (int*)(pin + in_bytes);
    meta_h[0] = len; ((unsigned char*)&meta_h[1])[0] = (unsigned char)scrambler_init;
    JRC_HIP(ctx, hipMemcpyAsync(ctx->scratch[0], pin, in_bytes + meta, hipMemcpyHostToDevice, ctx->stream));
    char* d0 = (char*)ctx->scratch[0];
    char* d1 = (char*)ctx->scratch[1];
    JRC_TRY(jrc_stream_encode_dev(ctx, mcs, n_data_carriers, 1, (const uint8_t*)d0, (long)in_bytes, (const int*)(d0 + in_bytes),
                                  (const uint8_t*)(d0 + in_bytes + 4), (jrc_cf32*)(d1 + 16), nsymb, (int*)d1, ctx->stream));
    JRC_HIP(ctx, hipMemcpyAsync(pin + in_bytes + meta, d1, out_bytes + 16, hipMemcpyDeviceToHost, ctx->stream));
    JRC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int produced = *(int*)(pin + in_bytes + meta);
    memcpy(out_symbols, pin + in_bytes + meta + 16, out_bytes);
    return produced;
}

extern "C" int jrc_stream_decode(jrc_ctx* ctx, int mcs, int n_data_carriers, int data_size_byte, const jrc_cf32* symbols, int n_symbols,
                                 uint8_t* out_payload, int* crc_ok)
{
    if (!ctx || !crc_ok) return JRC_ERR_INVALID_ARG;
    McsParams p;
    *crc_ok = 0;
    if (!mcs_params(mcs, n_data_carriers, p) || data_size_byte < 0 || data_size_byte > CODEC_MAX_PAYLOAD ||
        n_sym_for(data_size_byte, p.n_dbps) > CODEC_MAX_SYM)
        return JRC_ERR_UNSUPPORTED;                                                         // frame refused (:133-146)
    const int need = n_sym_for(data_size_byte, p.n_dbps) * n_data_carriers;
    if (!symbols || n_symbols < need) return jrc_fail(ctx, JRC_ERR_SHORT_INPUT, "stream_decoder: %d symbols given, %d needed", n_symbols, need);
    if (data_size_byte > 4 && !out_payload) return JRC_ERR_INVALID_ARG;
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t sb = sizeof(float2) * (size_t)need, pb = ((size_t)data_size_byte + 15) & ~(size_t)15;
    JRC_TRY(jrc_ensure_pinned(ctx, sb + 16 + pb + 16));
    JRC_TRY(jrc_ensure_scratch(ctx, 0, sb + 16));
    JRC_TRY(jrc_ensure_scratch(ctx, 1, pb + 16));
    char* pin = (char*)ctx->pinned;
    int* meta = (int*)pin;
    meta[0] = mcs; meta[1] = data_size_byte;
    memcpy(pin + 16, symbols, sb);
    JRC_HIP(ctx, hipMemcpyAsync(ctx->scratch[0], pin, sb + 16, hipMemcpyHostToDevice, ctx->stream));
    char* d0 = (char*)ctx->scratch[0];
    char* d1 = (char*)ctx->scratch[1];
    JRC_TRY(jrc_stream_decode_dev(ctx, n_data_carriers, 1, (const jrc_cf32*)(d0 + 16), need, (const int*)d0, (const int*)(d0 + 4),
                                  (uint8_t*)(d1 + 16), (long)pb, (int*)d1, ctx->stream));
    JRC_HIP(ctx, hipMemcpyAsync(pin + sb + 16, d1, pb + 16, hipMemcpyDeviceToHost, ctx->stream));
    JRC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int st = *(int*)(pin + sb + 16);
    if (st < 0) return JRC_ERR_UNSUPPORTED;
    *crc_ok = st;
    if (data_size_byte > 4) memcpy(out_payload, pin + sb + 16 + 16, (size_t)data_size_byte - 4);
    return data_size_byte > 4 ? data_size_byte - 4 : 0;
}
