#!/bin/bash
# Round 4: step-level parity under the non-default settings of this round's switches (tools/gpu/switch_matrix.sh runs the step tests per setting)
bash tools/gpu/switch_matrix.sh "PICONS_SPLIT=0" "PICONS_SPLIT_WGRAD=0" "PICONS_SPLIT_SPECTRAL=0" "PICONS_X6_TAIL_SPLIT=0" "PICONS_LANES=1" "PICONS_WINO=0" "PICONS_SPLIT_WGRAD=0 PICONS_SPECTRAL=0 PICONS_TAIL6=0"
