"""Drop-in shim: `import cuda_ext` (what /root/reference/model.py:12 and generator.py do) resolves to the
MI355X implementation.  Put this repository's root on PYTHONPATH ahead of the reference checkout."""
from exllama_amd.cuda_ext import *          # noqa: F401,F403
from exllama_amd.cuda_ext import exllama_ext, none_tensor  # noqa: F401
