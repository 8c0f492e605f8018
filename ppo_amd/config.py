"""Run configuration — host mirror of the reference's rl/config.py for the flags the PPO hot path reads.

Same flag names, defaults and access pattern as the reference (`rl.config.args`, a global
singleton read at call time; grouped flags appear as `--<prefix>_<name>` and are read as
`args.<prefix>.<name>`, rl/config.py:38-185).  Defaults cite rl/config.py line numbers.
Flags of out-of-scope subsystems (distillation, RND, replay, hashing, ...) are
accepted and ignored with a note, so existing launch lines keep working.
"""
import argparse
import sys


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


class _Group:
    """A prefixed flag group: fields declared as (name, type, default, help)."""
    FIELDS = ()

    def __init__(self, prefix):
        self._prefix = prefix
        for name, _t, default, _h in self.FIELDS:
            setattr(self, name, default)

    def add(self, parser):
        for name, t, default, h in self.FIELDS:
            kw = dict(type=str2bool, nargs="?", const=True) if t is bool else dict(type=t)
            parser.add_argument(f"--{self._prefix}_{name}", default=default, help=h, **kw)

    def update(self, ns):
        for name, *_ in self.FIELDS:
            setattr(self, name, getattr(ns, f"{self._prefix}_{name}"))

    def flatten(self):
        return {f"{self._prefix}_{name}": getattr(self, name) for name, *_ in self.FIELDS}


class OptimizerConfig(_Group):  # rl/config.py:249-311
    EPOCH_DEFAULTS = {"policy_opt": 2, "value_opt": 1, "distil_opt": 2}  # :261-267

    def __init__(self, prefix):
        super().__init__(prefix)
        self.epochs = self.EPOCH_DEFAULTS.get(prefix, 0)

    def add(self, parser):
        super().add(parser)
        parser.set_defaults(**{f"{self._prefix}_epochs": self.EPOCH_DEFAULTS.get(self._prefix, 0)})

    FIELDS = (
        ("optimizer", str, "adam", "[adam]"),
        ("epochs", int, 2, "training epochs per batch (default: policy 2, value 1, distil 2)"),
        ("mini_batch_size", int, 256, "examples per optimisation step (global, across ranks)"),
        ("lr", float, 2.5e-4, "learning rate"),
        ("lr_anneal", bool, False, "anneal learning rate linearly to 0"),
        ("adam_epsilon", float, 1e-5, "Adam epsilon"),
        ("adam_beta1", float, 0.9, "Adam beta1"),
        ("adam_beta2", float, 0.999, "Adam beta2"),
    )


class ModelConfig(_Group):  # rl/config.py:458-476
    FIELDS = (
        ("architecture", str, "dual", "[dual|single]  (north-star PPO = single)"),
        ("encoder", str, "nature", "[impala|mlp]  (nature has no HIP path)"),
        ("encoder_args", str, None, "dict of encoder arguments"),
        ("hidden_units", int, 256, "encoder output features"),
        ("head_scale", float, 0.1, "orthogonal-init gain of the heads"),
        ("head_bias", bool, True, "bias on the output heads"),
    )


class EnvConfig(_Group):  # rl/config.py:495-603
    FIELDS = (
        ("name", str, "Pong", "environment name"),
        ("type", str, "atari", "[atari|procgen|mujoco|classic|synthetic]"),
        ("embed_time", bool, True, "append a time channel"),
        ("embed_action", bool, True, "embed last action"),
        ("reward_normalization", str, "rms", "[off|rms]"),
        ("reward_normalization_clipping", float, 10.0, "clip rewards after normalisation, negative to disable (:509)"),
        ("max_repeated_actions", int, 100, "penalise repeating one action more than this many times (:513)"),
        ("repeated_action_penalty", float, 0.0, "the penalty (:514)"),
        ("warmup_period", int, 250, "random warm-up steps to desynchronise envs"),
        ("timeout", str, "auto", "episode step limit in agent steps: auto (atari 27000, procgen 1000, mujoco 1001, else off) or a number, 0 = off (:503)"),
        ("repeat_action_probability", float, 0.0, "ALE sticky actions (:504)"),
        ("noop_duration", int, 30, "maximum number of no-ops after a reset, 0 = none (:505)"),
        ("per_step_termination_probability", float, 0.0, "probability that a step ends the episode (:506)"),
        ("reward_clipping", str, "off", "[off|<R>|sqrt] (:507)"),
        ("deferred_rewards", int, 0, "pay all rewards at this step (-1: at the end), 0 = off (:510)"),
        ("full_action_space", bool, False, "ALE full action set (:517)"),
        ("resolution", str, "nature", "[full|nature|half|muzero] (:520)"),
        ("color_mode", str, "default", "[default|bw|rgb|yuv|hsv]; default = bw for atari, yuv for procgen (:521)"),
        ("frame_stack", int, -1, "frames stacked, -1 = 4 for atari else 1 (:522)"),
        ("frame_skip", int, -1, "simulator frames per agent step, -1 = 4 for atari else 1 (:523)"),
        ("embed_state", bool, False, "draw a compressed state history onto the frame (:526)"),
        ("atari_terminal_on_loss_of_life", bool, False, "(:529)"),
        ("atari_rom_check", bool, True, "(:530; the ROM table is not part of this build: accepted and ignored)"),
        ("procgen_difficulty", str, "hard", "[hard|easy] (:533)"),
        ("zero_obs", bool, False, "blank observations (the reference's --debug_zero_obs)"),
        ("synthetic_done_prob", float, 0.01, "synthetic env: per-step termination probability"),
        ("synthetic_threads", int, 16, "synthetic env: host threads generating observations (a GPU's share of the host cores)"),
        ("synthetic_actions", int, 0, "synthetic env: size of the action set (0 = 6, the Pong-shaped default)"),
        ("synthetic_shape", str, None, "synthetic env: observation shape 'C,H,W' (default 4,84,84; procgen-shaped: 3,64,64)"),
        ("pipeline_parts", int, 2, "split the envs into this many groups so host stepping overlaps the GPU policy step"),
        ("pipeline_leaves", int, 1, "step and upload each group in this many pieces, a piece's upload running while the next is stepped (measured: 0.494 / 0.558 / 0.792 ms per env step for 1 / 2 / 4 - a thread-pool dispatch per piece costs more than the overlap returns)"),
    )


class DistilConfig(_Group):  # rl/config.py:329-354
    FIELDS = (
        ("order", str, "after_policy", "[after_policy|before_policy]"),
        ("beta", float, 1.0, "weight of the policy constraint"),
        ("target", str, "value", "[value]  (return / advantage targets are not built)"),
        ("batch_size", int, -1, "distil batch size, negative = the rollout"),
        ("period", int, 1, "distil every this many batches"),
        ("loss", str, "kl_policy", "[kl_policy]"),
        ("max_heads", int, -1, "max TVF heads to distil, -1 = all"),
        ("force_ext", bool, False, "distil the ext value head even when TVF is on"),
        ("value_loss", str, "mse", "[mse]"),
        ("delay", float, 0, "millions of steps before distillation starts"),
        ("use_policy_opt", bool, False, "share the policy optimiser's moments"),
    )


class TVFConfig(_Group):  # rl/config.py:209-246
    FIELDS = (
        ("enabled", bool, False, "truncated value functions"),
        ("value_heads", int, 128, "number of horizon heads"),
        ("max_horizon", int, 30000, "longest horizon"),
        ("gamma", float, None, "TVF discount (None = gamma)"),
        ("coef", float, 1.0, "TVF loss coefficient"),
        ("head_spacing", str, "geometric", "[geometric|linear]"),
        ("return_mode", str, "advanced", "[standard|advanced|full|...]"),
        ("return_distribution", str, "exponential", "[fixed|exponential|uniform|hyperbolic|quadratic]"),
        ("return_samples", int, 8, "n-step samples per horizon"),
        ("return_use_log_interpolation", bool, False, "interpolate in log-horizon space"),
        ("include_ext", bool, False, "also train the ext value head in the value phase"),
        ("trimming", str, "off", "[off|timelimit|est_term] reduce horizons past the episode end to the time left (:217)"),
        ("trimming_mode", str, "average", "[interpolate|average|substitute|random] (:218)"),
        ("trim_advantages", str, "trimmed", "[trimmed|untrimmed|average] value estimate used for advantages (:219)"),
        ("trim_clip", float, -1.0, "if >= 0 clips how much trimming can change a value estimate (:220)"),
        ("eta_minh", int, 128, "estimated-termination trimming: minimum horizon (:221)"),
        ("eta_buffer", int, 32, "estimated-termination trimming: steps added to the percentile (:222)"),
        ("eta_percentile", float, 90.0, "estimated-termination trimming: percentile of episode lengths (:223)"),
        ("head_weighting", str, "off", "[off|h_weighted]"),
        ("horizon_dropout", float, 0.0, "fraction of horizons excluded per sample in the TVF loss (:224)"),
        ("feature_window", int, -1, "limits each head to a window of this many features (:233)"),
        ("feature_sparsity", float, 0.0, "zeros out this proportion of features for each head (:234)"),
    )


class Config:
    def __init__(self):
        self.policy_opt = OptimizerConfig("policy_opt")
        self.value_opt = OptimizerConfig("value_opt")
        self.distil_opt = OptimizerConfig("distil_opt")
        self.distil = DistilConfig("distil")
        self.model = ModelConfig("model")
        self.env = EnvConfig("env")
        self.tvf = TVFConfig("tvf")
        self._groups = (self.policy_opt, self.value_opt, self.distil_opt, self.distil, self.model, self.env, self.tvf)
        # top-level defaults (rl/config.py line numbers)
        self.agents = 256              # :791
        self.n_steps = 256             # :790
        self.gamma = 0.999             # :769
        self.lambda_policy = 0.95      # :772
        self.lambda_value = 0.95       # :773
        self.ppo_epsilon = 0.2         # :789
        self.entropy_bonus = 0.01      # :785
        self.ppo_vf_coef = 0.5         # :784
        self.max_grad_norm = 20.0      # :775
        self.grad_clip_mode = "global_norm"  # :776
        self.advantage_epsilon = 1e-8  # :792
        self.advantage_clipping = None
        self.ppo_epsilon_anneal = False
        self.entropy_scaling = "off"           # [off|average|uniform]
        self.entropy_scaling_base_actions = 18
        self.entropy_anneal = False
        self.advantage_epsilon_anneal_factor = 0.0
        self.anneal_target_epoch = None
        self.max_micro_batch_size = 512  # :760
        self.device = "cpu"            # :731 (the reference default; this build requires a GPU)
        self.upload_batch = False      # :732
        self.observation_normalization = False  # :756
        self.observation_normalization_epsilon = 0.003  # :757
        self.freeze_observation_normalization = False  # :758
        self.observation_scaling = "scaled"     # :755
        self.seed = -1                 # :749
        self.epochs = 50.0             # millions of env steps
        self.limit_epochs = None
        self.benchmark_mode = False
        self.disable_logging = False   # :733
        self.disable_ev = False
        self.output_folder = "./"
        self.experiment_name = "Run"
        self.run_name = "run"
        self.restore = "auto"          # :720
        self.initial_model = None      # :728
        self.checkpoint_every = int(10e6)  # :752
        self.save_checkpoints = True   # :736
        self.save_initial_checkpoint = False  # :737
        self.save_early_checkpoint = False    # :738
        self.debug_print_freq = 60     # DebugConfig.print_freq
        self.checkpoint_compression = True  # :726
        self.workers = -1              # :722
        self.threads = 2               # :723
        self.precision = "medium"      # :764
        self.use_intrinsic_rewards = False
        self.sync_envs = False
        self.override_reward_normalization_gamma = None  # :780
        self.log_folder = None
        self._ignored = []

    # ---- properties the reference derives (rl/config.py:885-901)
    RESOLUTIONS = {"full": (210, 160), "procgen": (64, 64), "nature": (84, 84), "muzero": (96, 96), "half": (105, 80)}

    @property
    def batch_size(self):
        return self.n_steps * self.agents

    @property
    def reward_normalization_gamma(self):  # rl/config.py:875-880
        if self.override_reward_normalization_gamma is not None:
            return self.override_reward_normalization_gamma
        return self.tvf.gamma if self.tvf.enabled else self.gamma

    @property
    def tvf_return_n_step(self):
        return round(1 / (1 - self.lambda_value)) if self.lambda_value < 1 else self.n_steps

    def build_parser(self):
        p = argparse.ArgumentParser(description="MI355X-native PPO trainer (drop-in for dremovd/PPO train.py)")
        a = p.add_argument
        a("--agents", type=int, default=self.agents)
        a("--n_steps", type=int, default=self.n_steps)
        a("--gamma", type=float, default=self.gamma)
        a("--lambda_policy", type=float, default=self.lambda_policy)
        a("--lambda_value", type=float, default=self.lambda_value)
        a("--ppo_epsilon", type=float, default=self.ppo_epsilon)
        a("--entropy_bonus", type=float, default=self.entropy_bonus)
        a("--ppo_vf_coef", type=float, default=self.ppo_vf_coef)
        a("--max_grad_norm", type=float, default=self.max_grad_norm)
        a("--grad_clip_mode", type=str, default=self.grad_clip_mode, help="[off|global_norm]")
        a("--advantage_epsilon", type=float, default=self.advantage_epsilon)
        a("--advantage_clipping", type=float, default=None)
        a("--ppo_epsilon_anneal", type=str2bool, nargs="?", const=True, default=False)
        a("--entropy_scaling", type=str, default="off", help="[off|average|uniform]")
        a("--entropy_scaling_base_actions", type=int, default=18)
        a("--entropy_anneal", type=str2bool, nargs="?", const=True, default=False)
        a("--advantage_epsilon_anneal_factor", type=float, default=0.0)
        a("--anneal_target_epoch", type=float, default=None)
        a("--max_micro_batch_size", type=int, default=self.max_micro_batch_size)
        a("--device", type=str, default=self.device)
        a("--upload_batch", type=str2bool, nargs="?", const=True, default=self.upload_batch)
        a("--observation_normalization", type=str2bool, nargs="?", const=True, default=False)
        a("--observation_normalization_epsilon", type=float, default=0.003)
        a("--freeze_observation_normalization", type=str2bool, nargs="?", const=True, default=False)
        a("--observation_scaling", type=str, default="scaled")
        a("--seed", type=int, default=self.seed)
        a("--epochs", type=float, default=self.epochs, help="millions of env steps to train for")
        a("--limit_epochs", type=float, default=None)
        a("--benchmark_mode", type=str2bool, nargs="?", const=True, default=False)
        a("--disable_logging", type=str2bool, nargs="?", const=True, default=False)
        a("--disable_ev", type=str2bool, nargs="?", const=True, default=False)
        a("--output_folder", type=str, default=self.output_folder)
        a("--experiment_name", type=str, default=self.experiment_name)
        a("--run_name", type=str, default=self.run_name)
        a("--restore", type=str, default=self.restore, help="[never|auto|always]")
        a("--initial_model", type=str, default=None, help="checkpoint (in log_folder) to start from at step 0")
        a("--log_folder", type=str, default=None)
        a("--checkpoint_every", type=int, default=self.checkpoint_every)
        a("--save_checkpoints", type=str2bool, nargs="?", const=True, default=True)
        a("--save_initial_checkpoint", type=str2bool, nargs="?", const=True, default=False)
        a("--save_early_checkpoint", type=str2bool, nargs="?", const=True, default=False)
        a("--debug_print_freq", type=int, default=self.debug_print_freq)
        a("--checkpoint_compression", type=str2bool, nargs="?", const=True, default=True)
        a("--workers", type=int, default=self.workers)
        a("--threads", type=int, default=self.threads)
        a("--precision", type=str, default=self.precision,
          help="[low|medium|high] (train.py:166-178): high = exact float32 everywhere; low / medium also allow the "
               "split-bf16 launches (3 bf16 MFMAs per product, ~16-bit products) where they exist - the residual blocks of "
               "the IMPALA encoder's 32-channel stacks")
        a("--use_intrinsic_rewards", type=str2bool, nargs="?", const=True, default=False)
        a("--sync_envs", type=str2bool, nargs="?", const=True, default=False)
        a("--override_reward_normalization_gamma", type=float, default=None)
        for g in self._groups:
            g.add(p)
        return p

    def setup(self, argv=None):
        """Parse argv (default sys.argv[1:]) like rl.config.args.setup() (rl/config.py:709-801)."""
        parser = self.build_parser()
        ns, unknown = parser.parse_known_args(sys.argv[1:] if argv is None else argv)
        for k, v in vars(ns).items():
            if any(k.startswith(g._prefix + "_") for g in self._groups):
                continue
            setattr(self, k, v)
        for g in self._groups:
            g.update(ns)
        self._ignored = [u for u in unknown if u.startswith("--")]
        self.verify()
        return self

    def verify(self):
        if self.model.architecture not in ("single", "dual"):
            raise ValueError(f"Invalid architecture {self.model.architecture}, use [dual|single]")
        if self.grad_clip_mode not in ("off", "global_norm"):
            raise ValueError("Invalid clip_mode.")
        if self.tvf.gamma is None:
            self.tvf.gamma = self.gamma
        # EnvConfig.auto (rl/config.py:563-600)
        if self.env.frame_skip in (None, -1):
            self.env.frame_skip = 4 if self.env.type == "atari" else 1
        if self.env.frame_stack in (None, -1):
            self.env.frame_stack = 4 if self.env.type == "atari" else 1
        if self.env.color_mode not in ("default", "bw", "rgb", "yuv", "hsv"):
            raise ValueError(f"Invalid color mode {self.env.color_mode}")
        if self.env.color_mode == "default":
            self.env.color_mode = {"atari": "bw", "procgen": "yuv"}.get(self.env.type, "bw")
        if self.env.timeout == "auto":
            if self.env.type == "atari":
                self.env.timeout = 27000
            elif self.env.type == "procgen":
                self.env.timeout = {"bigfish": 6000, "bossfight": 8000, "plunder": 4000}.get(self.env.name, 1000)
            elif self.env.type == "mujoco":
                self.env.timeout = (50 if self.env.name.lower() == "reacher" else 1000) + 1
            else:
                self.env.timeout = 0  # unlimited
        else:
            self.env.timeout = int(self.env.timeout)
        if self.env.type in ("procgen", "mujoco") and (self.env.frame_stack != 1 or self.env.frame_skip != 1):
            raise ValueError(f"Frame stacking / skipping not supported on {self.env.type} yet")  # (:555-560)
        if self.restore in ("True", "true", True):  # rl/config.py:810-812
            self.restore = "always"
        if self.restore not in ("always", "never", "auto"):
            raise ValueError(f"Expecting {self.restore} to be one of ['always', 'never', 'auto']")

    def flatten(self):
        d = {k: v for k, v in vars(self).items() if not k.startswith("_") and not isinstance(v, _Group)}
        for g in self._groups:
            d.update(g.flatten())
        return d


args = Config()
