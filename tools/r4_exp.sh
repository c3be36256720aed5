#!/bin/bash
for e in 0 1 2; do echo "== S3R_FIN_EXP=$e"; S3R_FIN_EXP=$e bash tools/r4_steptrace.sh | grep -E "wino2s_finish|wino2p_finish"; done
