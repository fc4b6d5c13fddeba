"""simple_knn._C of the reference (submodules/simple-knn/ext.cpp:15-17) exports one function."""
from online_lang_splatting_amd.simple_knn import distCUDA2  # noqa: F401
