"""Drop-in package name of the reference (`rl.*`): every module here re-exports the MI355X-native
implementation in `ppo_amd` under the reference's module path, so `import rl.rollout; rl.rollout.Runner`,
`rl.returns.gae`, `rl.config.args`, `rl.ppo.train`, `rl.models.TVFModel` resolve as they do in the
reference tree."""
