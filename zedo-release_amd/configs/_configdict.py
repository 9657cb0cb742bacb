"""Attribute-style config container.  The reference uses ml_collections.ConfigDict (configs/*.py); that
package is not installed offline, so this minimal stand-in is used when the import fails."""
try:
    from ml_collections import ConfigDict  # noqa: F401
except Exception:  # pragma: no cover - depends on the environment
    class ConfigDict(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError as e:
                raise AttributeError(k) from e

        def __setattr__(self, k, v):
            self[k] = v

        def to_dict(self):
            return {k: (v.to_dict() if isinstance(v, ConfigDict) else v) for k, v in self.items()}
