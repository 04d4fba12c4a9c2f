"""Import shim so the read-only reference can be imported in the survey
container, where torchaudio is not installed.  Only
``torchaudio.functional.melscale_fbanks`` is used by the reference
(speechcatcher/model/frontend/stft_frontend.py:73-81)."""
from . import functional  # noqa: F401
