// radar_kernels.h — launch interfaces shared between the per-block entry points and the fused chain
#pragma once
#include "jrc_internal.h"

// geometry of the frame buffer the channel-estimate kernel reads: element (f, port, item, sc) is at
// frames[f*frame_stride + port*port_stride + item*N + sc]  (strides in cf32 elements)
struct ChanestGeom {
    int  N, S;
    long frame_stride, port_stride;
    int  tx_item0, rx_item0;   // first item used on the TX / RX ports (N_pre [+ discard])
    int  interleave;
};

int launch_radar_chanest(jrc_ctx* ctx, int T, int R, const float2* d_frames, float2* d_H,
                         const ChanestGeom& g, int n_frames, hipStream_t stream);

// A6 + A7 + A1 fused (radar.hip): time-domain RX streams in, channel estimate out
struct DemodGeom {
    int  N, cp, S, R, logn;
    long tx_frame_stride, tx_port_stride;      // cf32 elements
    long rx_frame_stride, rx_stream_stride;    // cf32 elements (time domain)
    int  tx_item0, rx_sym0;                    // first symbol used on the TX / RX side
    int  interleave;
    int  blocks_per_frame;                     // > 1: a frame's receivers span several workgroups (kept on one XCD)
    int  n_xcd;                                // filled in by launch_demod_chanest from the context
};
bool demod_chanest_supported(int N, int T);
int launch_demod_chanest(jrc_ctx* ctx, int T, const float2* d_tx, const float2* d_rx_td, float2* d_H, DemodGeom g, int n_frames,
                         hipStream_t stream);

// generic batched power-of-two FFT with gr::fft::fft_vcc semantics (fft.hip)
int launch_fft_vcc(jrc_ctx* ctx, int n, int forward, int shift, const float* d_window, size_t batch,
                   const float2* d_in, float2* d_out, long in_stride, int in_offset, hipStream_t stream);

// estimator pieces (estimator.hip)
struct RaParams {
    int   vlen, n_inputs, n_range_bins, n_angle_bins;
    float noise_discard_range_m, noise_discard_angle_deg;
};
int launch_ra_partial(jrc_ctx* ctx, const float2* d_map, size_t total, PeakPartial* d_partials, int n_blocks,
                      hipStream_t stream);
int launch_ra_finalize(jrc_ctx* ctx, const float2* d_map, size_t map_stride, const PeakPartial* d_partials,
                       int partials_per_frame, const RaParams& prm, const float* d_range_bins,
                       const float* d_angle_bins, jrc_ra_result* d_results, int n_frames, int win_rows /* 0: d_map is the full map */,
                       hipStream_t stream);
void ra_finish_host(jrc_ra_result* r, float snr_threshold, float power_threshold);
