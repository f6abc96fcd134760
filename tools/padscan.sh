for pad in 0 32 160 544 1056 2080 4128 8224 16416 32800 65568; do echo "pad=$pad"; python tools/kbench.py --ldq-pad $pad --rpls 8,16 2>/dev/null | grep rpl; done
