# fp32 leg of the default bench run vs the stand-alone fp32 run: critical-path composition (scripts/rocpd_timeline.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kl; rocprofv3 --kernel-trace -d /tmp/kl -o kl -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > /tmp/leg.json 2>/dev/null
python3 $R/scripts/rocpd_timeline.py $(find /tmp/kl -name "*.db" | head -1) 12 2 vit_attn_f32s_kernel | head -24
rm -rf /tmp/kl; rocprofv3 --kernel-trace -d /tmp/kl -o kl -- python3 $R/bench.py --dtype fp32 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > /tmp/leg.json 2>/dev/null
python3 $R/scripts/rocpd_timeline.py $(find /tmp/kl -name "*.db" | head -1) 12 2 vit_attn_f32s_kernel | head -24
