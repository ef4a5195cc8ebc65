"""Minimal stand-in for `loguru` so the read-only reference under /root/reference imports in this
container (the real package is not installed and there is no network). Test infrastructure only."""


class _Logger:
    def __getattr__(self, name):
        def _noop(*args, **kwargs):
            return None

        return _noop


logger = _Logger()
