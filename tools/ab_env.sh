# A/B of environment switches on the GPU box:  bash tools/ab_env.sh [-n STEPS] VAR=a VAR=b ...   (one bench.py run per argument)
N=30
if [ "$1" = "-n" ]; then N=$2; shift 2; fi
for v in "$@"; do env $v python bench.py --steps $N --warmup 8 --cpu-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('$v', d['value'], d['ms_per_step'], d['median_ms_per_step'], r['frac'], {k:round(x,2) for k,x in r.get('traced_kernel_ms_per_step',{}).items()})"; done
