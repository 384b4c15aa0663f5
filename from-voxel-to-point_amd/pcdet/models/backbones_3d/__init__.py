"""See pcdet/models/__init__.py."""
