for v in default g256_pp g256s_nt default; do
  if [ $v = default ]; then unset X2V_LIB_PATH; else export X2V_LIB_PATH=$PWD/tools/probes/ab/$v/libx2v_hip.so; fi
  echo -n "$v: "; timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; a=80*(r['avg_launch_ms']+r['cross_attention']['avg_ms']); print('ms_per_step %.1f attention %.1f rest %.1f' % (d['ms_per_step'], a, d['ms_per_step']-a))"
done
