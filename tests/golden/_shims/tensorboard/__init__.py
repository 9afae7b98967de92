"""Import-time names only (reference misc.py:18-22); logging is out of scope."""
