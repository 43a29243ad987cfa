"""Select the diagnostic library (libmme_diag.so, `python -m multimodal_embeddings_amd.build --diag`) for a measurement
tool: the only build that reads the experiment switches of DESIGN.md 4.5 from the environment.  Import BEFORE
multimodal_embeddings_amd._lib.  The product library ignores those variables, so a tool that sets them without this
would silently measure the default configuration."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG = os.path.join(ROOT, "multimodal_embeddings_amd", "libmme_diag.so")


def use_diag_library(required: bool = True) -> bool:
    if os.path.exists(DIAG):
        os.environ["MME_LIB_PATH"] = DIAG
        os.environ["MME_ALLOW_LIB_OVERRIDE"] = "1"  # _lib.load_library refuses MME_LIB_PATH without this opt-in
        return True
    if required:
        sys.exit(f"{DIAG} is missing: build it first (python -m multimodal_embeddings_amd.build --diag)")
    return False
