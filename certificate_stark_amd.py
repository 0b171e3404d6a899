"""Import alias: the package directory is `certificate-stark_amd/` (not a valid Python identifier), so
`import certificate_stark_amd` resolves to it by turning this module into a package over that directory."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "certificate-stark_amd")]
__package__ = __name__
__spec__.submodule_search_locations = __path__
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
