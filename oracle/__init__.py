"""CPU oracle for the NPP-Net hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product (``npp_amd``) never does: it fails
loudly when ``libnpp_hip.so`` is missing instead of falling back to this code.

Parity status: PINNED for the embedder, MLP forward/backward, snake, adaptive
robust pixel loss, Adam/LR schedule, contextual-loss core, LPIPS head and the
patch sampler, against golden vectors generated in the build container by
importing the reference's own Python modules (``tests/golden/make_golden.py``).
UNPINNED beyond the feature-tensor boundary for the VGG16/VGG19 trunks: the
reference loads torchvision pretrained weights that are not under
``/root/reference`` and no reference test pins their outputs.
"""
from .npp_oracle import *  # noqa: F401,F403
from .npp_patch_oracle import *  # noqa: F401,F403
from .npp_light_oracle import *  # noqa: F401,F403
from .npp_search_oracle import *  # noqa: F401,F403
