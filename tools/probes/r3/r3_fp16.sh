cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -x -q -k "fp16 or bf16 or _h_ or halo or frames or h2 or cast or attention_h" 2>&1 | tail -15
python -m pytest tests/test_video_gpu.py -x -q -k "bf16_storage_full or highres or fused_groupnorm or bitwise" 2>&1 | tail -15
python tools/probes/r3/r3_sampler.py bf16:16:50 fp16:16:50 2>&1 | grep -v amdgpu
