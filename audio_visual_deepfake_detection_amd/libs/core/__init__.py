from .config import load_default_config, load_config, DEFAULTS

__all__ = ['load_default_config', 'load_config', 'DEFAULTS']
