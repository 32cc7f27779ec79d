"""Command-line tools mirroring the reference's tools/ (compress, decompress, datalist drivers) on the HIP path."""
