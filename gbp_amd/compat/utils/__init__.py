"""Host-side helpers under the reference's `utils` import names."""
from . import derivatives, gaussian, lie_algebra, read_balfile, transformations

__all__ = ['derivatives', 'gaussian', 'lie_algebra', 'read_balfile', 'transformations']
