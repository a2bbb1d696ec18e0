"""Registration policy of the compat registries: names the backend implements are protected."""
from openpvsg_amd import registry as R


class ProtectedRegistry(R.Registry):
    """Registry whose backend-provided entries cannot be displaced by later `register_module` calls
    (the reference's in-repo classes re-register the same names at import time)."""

    def __init__(self, base):
        super().__init__(base.name)
        self._base = base
        self._module_dict = base._module_dict           # shared storage with the backend registry
        self.protected = set(base._module_dict)

    def _register(self, cls, name=None, force=False):
        names = [name] if isinstance(name, str) else (name or [cls.__name__])
        for n in names:
            if n in self.protected:
                continue                                 # backend implementation wins
            self._module_dict[n] = cls


def protect(base):
    return ProtectedRegistry(base)


def _training_only(name):
    def f(*a, **k):
        raise NotImplementedError('%s is training-only; the MI355X backend covers inference (SURVEY.md section 2 #13)' % name)
    f.__name__ = name
    return f
