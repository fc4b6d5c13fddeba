"""Drop-in name for the reference's `simple_knn` extension (submodules/simple-knn): `from simple_knn._C import
distCUDA2` resolves to the MI355X implementation in online_lang_splatting_amd."""
