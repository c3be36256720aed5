#!/bin/bash
O=gpurun_out/r3n; mkdir -p $O
python tools/layer_bench.py --tiles 0,1,2,3,4,5,6,7 --variants 0 --rounds 5 > $O/lb_all_tiles.txt 2>&1
grep BEST $O/lb_all_tiles.txt; grep -v BEST $O/lb_all_tiles.txt | cut -c1-1100
