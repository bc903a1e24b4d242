for f in ecfft_amd/libecfft_hip.so ecfft_amd/variants/*.so; do
  echo "== $f"; ECFFT_LIB=$PWD/$f timeout 120 python bench.py --steps 5 --warmup 2 --cpu-log-n 0 --no-profile "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'ms', '%.3e'%d['value'])"
done
