from . import decode_codes, generate_codes  # noqa: F401
from .decode_codes import decode, get_codes  # noqa: F401
