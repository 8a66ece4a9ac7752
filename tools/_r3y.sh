for i in 1 2; do for cfg in 1 0; do echo "attn fork $cfg"; LAB_SKIP_LINEAR=1 DFINE_ATTN_BWD_FORK=$cfg python tools/linear_attn_bench.py 2>/dev/null | tail -2; done; done
