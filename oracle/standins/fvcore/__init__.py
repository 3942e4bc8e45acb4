"""Stand-in for the `fvcore` package (absent in this image).

TEST INFRASTRUCTURE ONLY. These modules exist so that `oracle/make_goldens.py`
can import the *unmodified* reference from /root/reference in the build
container and record golden vectors. They carry no arithmetic of the hot path
and are never imported by the product package.
"""
