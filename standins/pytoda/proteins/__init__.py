"""Placeholder for `pytoda.proteins.ProteinLanguage` (an isinstance target in
the reference's utils/utils.py:321; no protein path exists in this build)."""


class ProteinLanguage:
    pass
