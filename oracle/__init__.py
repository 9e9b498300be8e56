"""ORACLE — test infrastructure only (see ref_modules.py). Never imported by the product package."""
