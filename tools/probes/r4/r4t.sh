# sensitivity of the x3 conv launches to workgroups in flight (split-K slots)
for s in 512 768 1024 1536; do echo "== V2A_CONV_SLOTS=$s"; V2A_CONV_SLOTS=$s python tools/probes/emu_probe.py 2>&1 | grep "conv_igemm_f32x3"; done
