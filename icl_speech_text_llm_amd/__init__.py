"""Import alias for the ``icl-speech-text-llm_amd/`` package directory.

The task-mandated directory name contains hyphens, which Python's ``import`` statement cannot
spell.  This shim points its ``__path__`` at that directory, so
``import icl_speech_text_llm_amd.models.model_factory`` resolves to
``icl-speech-text-llm_amd/models/model_factory.py``.  No code lives here.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "icl-speech-text-llm_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f
