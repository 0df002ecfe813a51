#!/bin/bash
# Developer probe: variants side by side on ONE box (box-to-box variance is +-5 %)
for q in 8 24 8 24; do echo "hw queues $q"; GPU_MAX_HW_QUEUES=$q python tools/reader_probe2.py 512,512 4 2>&1 | grep "P="; done
