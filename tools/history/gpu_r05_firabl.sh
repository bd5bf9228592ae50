#!/bin/bash
# round 5: where k_fir_mfma's time goes — timing-only ablations (variant builds: for a in 1 2 3; do tools/variants.sh fmabl$a "-DTWX_FM_ABL=$a"; done)
# against the product's matrix-core form and the vector forms; profiles/r05_fir_mfma.txt
export TWX_FIR_MFMA=1
for v in "" fmabl1 fmabl2 fmabl3; do
  lib=""; [ -n "$v" ] && lib=amaranth_twstft_amd/variants/lib_$v.so
  [ -n "$v" ] && [ ! -f "$lib" ] && continue
  echo "== ${v:-matrix-core form}"; TWX_LIB=$lib python3 tools/aux_rates.py fir 2>/dev/null | grep '"kernel": "k_fir' | head -1 | cut -c1-200
done
echo "== vector forms (TWX_FIR_MFMA=0)"; TWX_FIR_MFMA=0 python3 tools/aux_rates.py fir 2>/dev/null | grep '"kernel": "k_fir' | head -1 | cut -c1-200
