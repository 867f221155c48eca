# runs ON THE GPU BOX: backward time vs the number of workgroups dX / dW get (-DBF_EXPERIMENT build as variants/exp.so)
cd $GRAFT_REPO_ROOT
export FASTNERF_LIB=$PWD/fast-learning-nerf_amd/variants/exp.so
python tools/time_bwd_parts.py
for w in 192 128 112 96 64; do FASTNERF_DW_WGS=$w python tools/time_bwd_parts.py; done
for w in 384 288 256 192; do FASTNERF_DX_WGS=$w python tools/time_bwd_parts.py; done
