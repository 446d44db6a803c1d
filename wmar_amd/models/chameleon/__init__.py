"""Host-side mirrors of the Chameleon inference pieces around the decode engine (deps/chameleon/inference)."""
from .vocab import VocabInfo, VocabTranslation  # noqa: F401
