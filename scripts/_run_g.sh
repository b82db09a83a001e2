for v in 0 4096 8192 16384; do echo "AGP_U1_F32_ABOVE=$v"; AGP_U1_F32_ABOVE=$v python scripts/time_mixed.py 32768 2>&1 | grep config; done
