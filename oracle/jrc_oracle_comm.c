/*
 * jrc_oracle_comm.c — CPU restatement (TEST INFRASTRUCTURE) of the comm-side rows of the hot path:
 * C1 mimo_ofdm_equalizer, C2 mimo_precoder, C3 steering.  PARITY UNPINNED (see jrc_oracle.h).
 */
#include "jrc_oracle.h"
