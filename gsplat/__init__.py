"""Import shim: with the repository root on sys.path, `import gsplat` (and
`from gsplat.project_gaussians_2d import ...`, as the reference's model files do) resolves to
gaussianimage_plus_amd.gsplat.  See INTEGRATION.md."""
import sys as _sys

import gaussianimage_plus_amd as _pkg

_pkg.install_as_gsplat()
_self = _sys.modules["gsplat"]
globals().update({k: getattr(_self, k) for k in dir(_self) if not k.startswith("__")})
