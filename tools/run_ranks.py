"""Rank entry point of `python -m npp_amd.run --gpus N` (torch.distributed.run needs a script path)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npp_amd.run import main  # noqa: E402

if __name__ == "__main__":
    raise SystemExit(main())
