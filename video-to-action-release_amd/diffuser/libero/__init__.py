"""Host glue of the joint loop (SURVEY.md section 8b, row `diffuser.libero.*`): the trainer that alternates video-guided rollouts
with policy train steps, its dataset / video-model / text-encoder factories and the evaluation harness -- the callers on either
side of the HIP hot path.  The Libero simulator itself is not part of this package: anything that implements the `env_list`
protocol of `lb_synthetic_env.SyntheticLiberoEnvList` can be plugged in."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)      # the rest of the package comes from the user's checkout
