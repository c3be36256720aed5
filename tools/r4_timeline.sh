#!/bin/bash
for L in "$@"; do S3R_LIB=tools/alt/abl.so S3R_ABL=7 python tools/timeline.py --layer $L 2>&1 | grep -v amdgpu; done
