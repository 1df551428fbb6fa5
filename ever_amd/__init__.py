"""ever_amd: MI355X-native engine behind EVer's ERModule / registry / config API."""
__version__ = '0.1.0'
