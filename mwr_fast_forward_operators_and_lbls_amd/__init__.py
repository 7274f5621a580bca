"""MI355X-native line-by-line microwave forward operator (HATPRO TBs from radiosonde profiles).

Drop-in for the reference's pyrtlib hot path only
(python_src/proc/PyRTlib_processing.py:83-197): see DESIGN.md.
"""
__version__ = "0.1.0"
