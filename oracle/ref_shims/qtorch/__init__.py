"""TEST INFRASTRUCTURE ONLY — stand-in for the `qtorch` package (absent offline) so that the reference's converter
(tools/convert/converter.py:13) imports.  See quant.py."""
