"""bench.py's legs (measurement code; nothing here is imported by the product)."""
