#!/bin/bash
# round 5: where k_fir_mfma's time goes — timing-only ablations (variant builds -DTWX_FM_ABL=n, tools/variants.sh) against the product
for v in "" fmnoswap; do
  lib=""; [ -n "$v" ] && lib=amaranth_twstft_amd/variants/lib_$v.so
  echo "== ${v:-product}"; TWX_LIB=$lib python3 tools/aux_rates.py fir 2>/dev/null | grep '"kernel": "k_fir' | head -1 | cut -c1-200
done
echo "== vector forms (TWX_FIR_MFMA=0)"; TWX_FIR_MFMA=0 python3 tools/aux_rates.py fir 2>/dev/null | grep '"kernel": "k_fir' | head -1 | cut -c1-200
