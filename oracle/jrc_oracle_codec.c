/*
 * jrc_oracle_codec.c — CPU restatement (TEST INFRASTRUCTURE, not product) of the bit codec around the comm chain
 * (SURVEY §8(f) rank 4): stream_encoder_impl::general_work (lib/stream_encoder_impl.cc:76-270), the helpers of
 * lib/utils.cc:26-290, stream_decoder_impl::decode / descramble (lib/stream_decoder_impl.cc:205-435) and the windowed
 * SSE2 Viterbi decoder (lib/viterbi_decoder.cc:62-330), whose __m128i byte arithmetic is restated lane by lane.
 * PARITY UNPINNED (see jrc_oracle.h).  Third-party pieces restated from published behaviour: boost::crc_32_type
 * (CRC-32/ISO-HDLC) and the gr::digital 3.8 constellation objects (bpsk, qpsk; 16qam from recollection of
 * gr-digital/lib/constellation.cc, level = sqrt(0.1), index bits {re>0, |re|<2 level, im>0, |im|<2 level}).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "jrc_oracle.h"

#define ORC_MAX_PAYLOAD 3100                                            /* lib/utils.h MAX_PAYLOAD_SIZE */
#define ORC_MAX_ENCODED_BITS ((16 + 8 * ORC_MAX_PAYLOAD + 6) * 2 + 288) /* lib/utils.h MAX_ENCODED_BITS */

uint32_t orc_crc32(const uint8_t* p, size_t n)      /* boost::crc_32_type: poly 0x04C11DB7 reflected, init/xorout ~0 */
{
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) {
        c ^= p[i];
        for (int b = 0; b < 8; b++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
    }
    return c ^ 0xFFFFFFFFu;
}

/* packet_param (lib/utils.cc:26-53) on top of orc_mcs_params */
static int packet_params(int mcs, int n_dc, int data_size_byte, int* n_bpsc, int* n_cbps, int* n_dbps, int* n_sym,
                         int* n_data_bits, int* n_pad_bits, int* n_encoded_bits)
{
    int rf;
    if (orc_mcs_params(mcs, n_dc, n_bpsc, n_cbps, n_dbps, &rf) != 0) return -1;
    *n_sym = (int)ceil((16 + 8 * data_size_byte + 6) / (double)*n_dbps);
    *n_data_bits = *n_sym * *n_dbps;
    *n_pad_bits = *n_data_bits - (16 + 8 * data_size_byte + 6);
    *n_encoded_bits = *n_sym * *n_cbps;
    return 0;
}

static int ones8(int n) { int s = 0; for (int i = 0; i < 8; i++) s += (n >> i) & 1; return s; }

void orc_constellation_point(int bpsc, int value, float* re, float* im)
{
    if (bpsc == 1) { *re = value ? 1.0f : -1.0f; *im = 0.0f; return; }                 /* constellation_bpsk */
    if (bpsc == 2) {                                                                   /* constellation_qpsk, /2 (:218-221) */
        const float s = 0.707107f;
        *re = ((value & 1) ? s : -s) / (float)2.0; *im = ((value & 2) ? s : -s) / (float)2.0;
        return;
    }
    const float level = sqrtf(0.1f);                                                   /* constellation_16qam */
    const float a = (value & 2) ? 1.0f : 3.0f, b = (value & 8) ? 1.0f : 3.0f;
    *re = ((value & 1) ? a : -a) * level; *im = ((value & 4) ? b : -b) * level;
}

int orc_constellation_decide(int bpsc, float re, float im)
{
    if (bpsc == 1) return re > 0;
    if (bpsc == 2) return 2 * (im > 0) + (re > 0);
    const float level = sqrtf(0.1f);
    return (re > 0) | ((fabsf(re) < 2 * level) << 1) | ((im > 0) << 2) | ((fabsf(im) < 2 * level) << 3);
}

/* stream_encoder: psdu[len] (first byte = packet type, as the block reads it) -> n_sym*n_dc complex symbols.
 * Returns the number of symbols, or -1 (packet too large, :139-143).  *pdu_len_tag = len + 4. */
int orc_stream_encode(int mcs, int n_dc, const uint8_t* psdu, int len, int scrambler_init, float* out_sym, int* n_ofdm_sym,
                      int* pdu_len_tag)
{
    int n_bpsc, n_cbps, n_dbps, n_sym, ndb, npad, neb;
    if (len + 4 > ORC_MAX_PAYLOAD) return -1;
    if (packet_params(mcs, n_dc, len + 4, &n_bpsc, &n_cbps, &n_dbps, &n_sym, &ndb, &npad, &neb) != 0) return -2;
    uint8_t* pkt = (uint8_t*)calloc((size_t)len + 4, 1);
    memcpy(pkt, psdu, (size_t)len);
    const uint32_t fcs = orc_crc32(psdu, (size_t)len);
    memcpy(pkt + len, &fcs, 4);                                                         /* :153-156, little endian */
    char* data_bits = (char*)calloc((size_t)ndb, 1);
    char* scrambled = (char*)calloc((size_t)ndb, 1);
    char* encoded = (char*)calloc((size_t)ndb * 2, 1);
    char* punct = (char*)calloc((size_t)neb + 8, 1);
    for (int i = 0; i < len + 4; i++)                                                   /* generate_bits (utils.cc:136-149) */
        for (int b = 0; b < 8; b++) data_bits[16 + i * 8 + b] = !!(pkt[i] & (1 << b));
    int state = (char)scrambler_init;                                                   /* scramble (:151-162) */
    for (int i = 0; i < ndb; i++) {
        const int feedback = (!!(state & 64)) ^ (!!(state & 8));
        scrambled[i] = (char)(feedback ^ data_bits[i]);
        state = ((state << 1) & 0x7e) | feedback;
    }
    memset(scrambled + ndb - npad - 6, 0, 6);                                           /* reset_tail_bits (:165-168) */
    state = 0;                                                                          /* convolutional_encoding (:183-193) */
    for (int i = 0; i < ndb; i++) {
        state = ((state << 1) & 0x7e) | scrambled[i];
        encoded[2 * i] = (char)(ones8(state & 0155) % 2);
        encoded[2 * i + 1] = (char)(ones8(state & 0117) % 2);
    }
    char* o = punct;                                                                    /* puncturing (:196-224) */
    for (int i = 0; i < ndb * 2; i++) {
        if (mcs == 0 || mcs == 2 || mcs == 4) *o++ = encoded[i];
        else { const int mod = i % 6; if (!(mod == 3 || mod == 4)) *o++ = encoded[i]; }
    }
    const int nsymb = n_sym * n_dc;                                                     /* split_symbols (:256-271) + mapping */
    const char* in = punct;
    for (int i = 0; i < nsymb; i++) {
        int v = 0;
        for (int k = 0; k < n_bpsc; k++) v |= (*in++) << k;
        orc_constellation_point(n_bpsc, v, &out_sym[2 * i], &out_sym[2 * i + 1]);
    }
    free(pkt); free(data_bits); free(scrambled); free(encoded); free(punct);
    *n_ofdm_sym = n_sym; *pdu_len_tag = len + 4;
    return nsymb;
}

/* ---- viterbi_decoder (lib/viterbi_decoder.cc), byte lanes of the __m128i registers written out ---- */
typedef struct {
    uint8_t metric[2][64], path[2][64];
    uint8_t mmresult[64], ppresult[24][64];
    uint8_t branchtab[2][32];
    int store_pos;
} vit_t;

static int parity8(int v) { return ones8(v & 0xff) & 1; }

static void vit_init(vit_t* v)                                                          /* :318-338 */
{
    memset(v, 0, sizeof(*v));
    const int polys[2] = {0x6d, 0x4f};
    for (int i = 0; i < 32; i++) {
        v->branchtab[0][i] = (uint8_t)parity8((2 * i) & polys[0]);
        v->branchtab[1][i] = (uint8_t)parity8((2 * i) & polys[1]);
    }
}

/* one trellis step = one of the two halves of viterbi_butterfly2_sse2 (:87-180): cur -> nxt */
static void vit_step(vit_t* v, int cur, const uint8_t* sym)
{
    const int nxt = cur ^ 1;
    for (int k = 0; k < 32; k++) {
        uint8_t metsvm, metsv;
        if (sym[0] == 2) { metsvm = v->branchtab[1][k] ^ sym[1]; metsv = (uint8_t)(1 - metsvm); }
        else if (sym[1] == 2) { metsvm = v->branchtab[0][k] ^ sym[0]; metsv = (uint8_t)(1 - metsvm); }
        else { metsvm = (uint8_t)((v->branchtab[0][k] ^ sym[0]) + (v->branchtab[1][k] ^ sym[1])); metsv = (uint8_t)(2 - metsvm); }
        const uint8_t m0 = (uint8_t)(v->metric[cur][k] + metsv), m1 = (uint8_t)(v->metric[cur][k + 32] + metsvm);
        const uint8_t m2 = (uint8_t)(v->metric[cur][k] + metsvm), m3 = (uint8_t)(v->metric[cur][k + 32] + metsv);
        const int d0 = (int8_t)(uint8_t)(m0 - m1) > 0, d1 = (int8_t)(uint8_t)(m2 - m3) > 0;     /* _mm_cmpgt_epi8(sub, 0) */
        const uint8_t shift0 = (uint8_t)(v->path[cur][k] << 1), shift1 = (uint8_t)((uint8_t)(v->path[cur][k + 32] << 1) + 1);
        v->metric[nxt][2 * k] = d0 ? m0 : m1;       v->path[nxt][2 * k] = d0 ? shift0 : shift1;
        v->metric[nxt][2 * k + 1] = d1 ? m2 : m3;   v->path[nxt][2 * k + 1] = d1 ? shift0 : shift1;
    }
}
/* note: _mm_slli_epi16 shifts 16-bit lanes; bit 7 of a path byte would spill into its neighbour, but paths are zeroed
 * every 8 steps (get_output) so bit 7 is never set when a shift happens — byte shifts are identical. */

static uint8_t vit_get_output(vit_t* v, int ntraceback)                                 /* :183-225, works on metric[0]/path[0] */
{
    v->store_pos = (v->store_pos + 1) % ntraceback;
    memcpy(v->mmresult, v->metric[0], 64);
    memcpy(v->ppresult[v->store_pos], v->path[0], 64);
    int beststate = 0, bestmetric = v->mmresult[0], minmetric = v->mmresult[0];
    for (int i = 1; i < 64; i++) {
        if (v->mmresult[i] > bestmetric) { bestmetric = v->mmresult[i]; beststate = i; }
        if (v->mmresult[i] < minmetric) minmetric = v->mmresult[i];
    }
    int pos = v->store_pos;
    for (int i = 0; i < ntraceback - 1; i++) {
        beststate = v->ppresult[pos][beststate] >> 2;
        pos = (pos - 1 + ntraceback) % ntraceback;
    }
    const uint8_t out = v->ppresult[pos][beststate];
    for (int i = 0; i < 64; i++) { v->path[0][i] = 0; v->metric[0][i] = (uint8_t)(v->metric[0][i] - (uint8_t)minmetric); }
    return out;
}

/* viterbi_decoder::decode (:258-291): in = n_sym*n_cbps hard bits; decoded gets >= n_data_bits bits (multiple of 8).
 * Symbols past the end of the frame read as 0 (a freshly constructed decoder; the reference reads whatever an earlier,
 * longer frame left in its buffers). */
int orc_viterbi_windowed(int mcs, int n_sym, int n_cbps, int n_data_bits, const uint8_t* in, uint8_t* decoded)
{
    const int half = (mcs == 0 || mcs == 2 || mcs == 4);
    const int ntraceback = half ? 5 : 10;
    static const uint8_t P34[6] = {1, 1, 1, 0, 0, 1};
    uint8_t* dep = (uint8_t*)calloc((size_t)ORC_MAX_ENCODED_BITS + 4096, 1);
    if (half) memcpy(dep, in, (size_t)n_sym * n_cbps);                                  /* :232-234 */
    else {                                                                              /* :236-252 */
        int count = 0;
        for (int i = 0; i < n_sym; i++)
            for (int k = 0; k < n_cbps; k++) {
                while (P34[count % 6] == 0) { dep[count] = 2; count++; }
                dep[count] = in[i * n_cbps + k]; count++;
                while (P34[count % 6] == 0) { dep[count] = 2; count++; }
            }
    }
    vit_t* v = (vit_t*)malloc(sizeof(vit_t));
    vit_init(v);
    int in_count = 0, out_count = 0, n_decoded = 0;
    while (n_decoded < n_data_bits) {
        if ((in_count % 4) == 0) {
            const uint8_t* s = &dep[in_count & 0xfffffffc];
            vit_step(v, 0, s);                      /* metric0 -> metric1 */
            vit_step(v, 1, s + 2);                  /* metric1 -> metric0 */
            if (in_count > 0 && (in_count % 16) == 8) {
                const uint8_t c = vit_get_output(v, ntraceback);
                if (out_count >= ntraceback) {
                    for (int i = 0; i < 8; i++) decoded[(out_count - ntraceback) * 8 + i] = (c >> (7 - i)) & 0x1;
                    n_decoded += 8;
                }
                out_count++;
            }
        }
        in_count++;
    }
    free(v); free(dep);
    return n_decoded;
}

/* stream_decoder: equalised symbols [n_sym][n_dc] -> payload.  out_payload gets data_size_byte - 4 bytes (the PSDU without
 * its CRC, what the block publishes at send_out_bytes + info_bytes); returns 1 if the CRC residue matches (:246), else 0;
 * -1 if the frame is refused (:133-146). */
int orc_stream_decode(int mcs, int n_dc, int data_size_byte, const float* sym, uint8_t* out_payload)
{
    int n_bpsc, n_cbps, n_dbps, n_sym, ndb, npad, neb;
    if (packet_params(mcs, n_dc, data_size_byte, &n_bpsc, &n_cbps, &n_dbps, &n_sym, &ndb, &npad, &neb) != 0) return -1;
    const int max_sym = ((16 + 8 * ORC_MAX_PAYLOAD + 6) / 24) + 1;
    if (!(n_sym <= max_sym && data_size_byte <= ORC_MAX_PAYLOAD)) return -1;
    uint8_t* bits = (uint8_t*)calloc((size_t)neb + 8, 1);
    for (int i = 0; i < n_sym * n_dc; i++) {                                            /* :166-169, :227-233 */
        const int d = orc_constellation_decide(n_bpsc, sym[2 * i], sym[2 * i + 1]);
        for (int k = 0; k < n_bpsc; k++) bits[i * n_bpsc + k] = !!(d & (1 << k));
    }
    uint8_t* dec = (uint8_t*)calloc((size_t)ndb + 64, 1);
    orc_viterbi_windowed(mcs, n_sym, n_cbps, ndb, bits, dec);
    uint8_t* ob = (uint8_t*)calloc((size_t)data_size_byte + 2 + 8, 1);                  /* descramble (:406-433) */
    int state = 0;
    for (int i = 0; i < 7; i++) if (dec[i]) state |= 1 << (6 - i);
    ob[0] = (uint8_t)state;
    for (int i = 7; i < data_size_byte * 8 + 16; i++) {
        const int feedback = (!!(state & 64)) ^ (!!(state & 8));
        const int bit = feedback ^ (dec[i] & 0x1);
        ob[i / 8] |= (uint8_t)(bit << (i % 8));
        state = ((state << 1) & 0x7e) | feedback;
    }
    const int ok = orc_crc32(ob + 2, (size_t)data_size_byte) == 558161692u;             /* :245-246 */
    if (data_size_byte > 4) memcpy(out_payload, ob + 2, (size_t)data_size_byte - 4);
    free(bits); free(dec); free(ob);
    return ok;
}
