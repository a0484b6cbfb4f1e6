for i in 1 2; do
for lib in "" nw4; do
  if [ -n "$lib" ]; then export ROMANHIP_LIB=$GRAFT_REPO_ROOT/romanimpreprocess_amd/libromanhip_$lib.so; else unset ROMANHIP_LIB; fi
  python3 bench.py --ipc-dtype f64 --no-cpu-baseline --no-extras --steps 15 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('f64 lib=${lib:-cur}', round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), d['roofline']['kernel_form'])"
done; done
