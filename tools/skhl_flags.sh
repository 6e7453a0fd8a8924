for f in 0 64 80 112; do echo "== DEER_SKHL_FLAGS=$f"; DEER_SKHL_FLAGS=$f python tools/bench_skinny_hl.py 112 2>&1 | grep -E "xa ff1|down|out " ; done
