from .smiles_language import SMILESLanguage  # noqa: F401
