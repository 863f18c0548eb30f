"""Namespace bridge only: this build has no ``mst.models.utils`` modules of its own.  ``functions``,
``transformer_blocks`` and ``rotary_embedding_torch`` (scripts/main_predict.py:30) resolve in the reference checkout that
follows on ``sys.path`` (see mst/__init__.py); the reference's eager ``from .functions import ...`` is left to the caller."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
