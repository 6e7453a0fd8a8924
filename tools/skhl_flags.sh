for f in 0 512 768 1280 ; do echo "== DEER_SKHL_FLAGS=$f (depth $((f/256)))"; DEER_SKHL_FLAGS=$f python tools/bench_skinny_hl.py 56 2>&1 | grep -E "xa ff1|down|out |7b down" ; done
