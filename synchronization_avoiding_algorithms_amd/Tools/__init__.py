"""Drop-in counterparts of the reference's ``Tools`` package for the explicit-dynamics hot path.

Same module and function names as /root/reference/Tools (``from Tools.X import *`` keeps working if this
package is put on the path as ``Tools``); the element loop, the update and the shared-node sum run on
the MI355X through ``libsaa_hip.so``.  Only what the path needs is present.
"""
