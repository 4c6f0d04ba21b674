from . import recipe  # noqa: F401
