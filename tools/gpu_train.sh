set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
python -m pytest tests/test_fullsize_gpu.py tests/test_backward_fir.py tests/test_backward_fast.py tests/test_modules.py -m gpu -x -q 2>&1 | tail -2
for k in combsub sins; do for r in 1 2; do python tools/train_step_probe.py $k 2>&1 | tail -1; DDSP_HIP_ONE_STREAM=1 python tools/train_step_probe.py $k 2>&1 | tail -1 | sed 's/$/  <- one stream/'; done; done
