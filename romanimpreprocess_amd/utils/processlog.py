"""Host-side processing log (string accumulator).

Same surface as the reference's ``utils/processlog.py:12-56``: ``output`` text,
``reffiles`` dict, ``append``.  Kernels never log; the host wrapper appends the
lines the reference would have written where they are cheap to produce.
"""


class ProcessLog:
    def __init__(self):
        self.output = ""
        self.reffiles = {}

    def append(self, newoutput):
        self.output += newoutput
