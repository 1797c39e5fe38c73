"""Import alias: `celebrity_image_denoiser_amd` -> the sources in `celebrity-image-denoiser_amd/`.

The project directory carries the upstream repo's hyphenated name, which is not a legal Python
identifier; this two-line package points its search path at that directory and runs its
`__init__`, so `import celebrity_image_denoiser_amd` (and its submodules) resolve there.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "celebrity-image-denoiser_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
