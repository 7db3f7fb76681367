/* -*- c++ -*- */
/* SWIG interface of the MI355X build of gr-mimo_ofdm_jrc for GNU Radio 3.8: the module and class names the reference's generated
 * flowgraph code imports (`import mimo_ofdm_jrc`; `mimo_ofdm_jrc.mimo_ofdm_radar(...)`), one GR_SWIG_BLOCK_MAGIC2 per block.
 * All blocks are declared in ONE header (host/jrc_blocks.h); blocks the reference has and this build does not (the Qt GUI sinks and
 * usrp_mimo_trx, SURVEY.md §2.1 rows 18-19: display and hardware I/O, out of scope) are not wrapped. */
#define MIMO_OFDM_JRC_API
#define JRC_WITH_GNURADIO

%include "gnuradio.i"

%{
#include "jrc_blocks.h"
%}

%include "jrc_blocks.h"

GR_SWIG_BLOCK_MAGIC2(mimo_ofdm_jrc, mimo_ofdm_radar);
GR_SWIG_BLOCK_MAGIC2(mimo_ofdm_jrc, radar_chain);
GR_SWIG_BLOCK_MAGIC2(mimo_ofdm_jrc, matrix_transpose);
GR_SWIG_BLOCK_MAGIC2(mimo_ofdm_jrc, range_angle_estimator);
GR_SWIG_BLOCK_MAGIC2(mimo_ofdm_jrc, ofdm_cyclic_prefix_remover);
GR_SWIG_BLOCK_MAGIC2(mimo_ofdm_jrc, fft_peak_detect);
GR_SWIG_BLOCK_MAGIC2(mimo_ofdm_jrc, mimo_ofdm_equalizer);
GR_SWIG_BLOCK_MAGIC2(mimo_ofdm_jrc, mimo_precoder);
GR_SWIG_BLOCK_MAGIC2(mimo_ofdm_jrc, target_simulator);
GR_SWIG_BLOCK_MAGIC2(mimo_ofdm_jrc, stream_encoder);
GR_SWIG_BLOCK_MAGIC2(mimo_ofdm_jrc, stream_decoder);
GR_SWIG_BLOCK_MAGIC2(mimo_ofdm_jrc, moving_avg);
GR_SWIG_BLOCK_MAGIC2(mimo_ofdm_jrc, ofdm_frame_generator);
GR_SWIG_BLOCK_MAGIC2(mimo_ofdm_jrc, zero_pad);
GR_SWIG_BLOCK_MAGIC2(mimo_ofdm_jrc, frame_detector);
GR_SWIG_BLOCK_MAGIC2(mimo_ofdm_jrc, frame_sync);
