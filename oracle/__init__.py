"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's hot path (map builders + cv2.remap fixed-point
gather + two-stage view synthesis).  See cv_remap_oracle.c, maps.py, cpu_ref.py.
Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
