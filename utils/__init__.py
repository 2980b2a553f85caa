"""Drop-in for the reference's `utils` package on the segmentation-network path:
`utils.model` keeps the names and argument meaning of /root/reference
utils/model.py; everything underneath runs on libmiseg.so (MI355X)."""
