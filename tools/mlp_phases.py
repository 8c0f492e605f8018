"""Times the fused MLP launches (Humanoid shape) for one build of the library (PPO_AMD_LIB): the inference forward at 128
rows and the training pair at 256.  tools/mlp_phases.sh runs it over the PPO_TUNE_MLP_STOP builds (results of those are
garbage by design: the kernel leaves early)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppo_amd import models  # noqa: E402

torch.manual_seed(0)
hz = list(range(128))
m = models.TVFModel("mlp", input_dims=(377,), actions=17, device="cuda", architecture="dual", hidden_units=256,
                    encoder_activation_fn="tanh", head_scale=0.1, head_bias=True, tvf_fixed_head_horizons=hz)
net = m.value_net
x128, x256 = torch.randn(128, 377, device="cuda"), torch.randn(256, 377, device="cuda")
ret, tvf_ret = torch.randn(256, 1, device="cuda"), torch.randn(256, 128, device="cuda")
w = torch.ones(128, device="cuda")


def timeit(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


net.use_plans = False
fwd = timeit(lambda: net.encode(x128, train=False))
fwd256 = timeit(lambda: net.encode(x256, train=False))
fwd16 = timeit(lambda: net.encode(x128[:16], train=False))
train128 = timeit(lambda: net.value_minibatch(x256[:128], returns=ret[:128], tvf_returns=tvf_ret[:128], tvf_weights=w))
train = timeit(lambda: net.value_minibatch(x256, returns=ret, tvf_returns=tvf_ret, tvf_weights=w))
step = timeit(lambda: (net.value_minibatch(x256, returns=ret, tvf_returns=tvf_ret, tvf_weights=w), net.adam_step()))
print(f"{os.path.basename(os.environ.get('PPO_AMD_LIB', 'default')):>26s}  forward[16/128/256] {fwd16:6.1f} {fwd:6.1f} {fwd256:6.1f} us   "
      f"train pair[128/256] {train128:6.1f} {train:6.1f} us   + adam {step:6.1f} us")
