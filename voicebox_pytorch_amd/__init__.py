"""Import shim: the product package lives in the directory `voicebox-pytorch_amd/` (a name Python
cannot import directly).  `import voicebox_pytorch_amd` resolves to it."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "voicebox-pytorch_amd")
__path__ = [_real]
_init = _os.path.join(_real, "__init__.py")
with open(_init) as _f:
    exec(compile(_f.read(), _init, "exec"))
