"""Drop-in for the reference's un-vendored `simple_knn` dependency (only `simple_knn._C.distCUDA2` is used:
/root/reference/scene/saro_gaussian.py:21, :187)."""
