"""log()/qlog() with the reference's behaviour (scripts/lib/logger.py:10-47): append to
<analysis>/messages-<hostname> once init() was called, echo to stdout unless quiet."""
import os
import socket
from datetime import datetime

logfile = None
logbuf = []


def init(analysis_path):
    global logfile
    logfile = os.path.join(analysis_path, "messages-" + socket.gethostname())


def log(*args, quiet=False, fancy=False):
    global logbuf
    stamp = str(datetime.now()) + ": "
    msg = " ".join(str(a) for a in args)
    if fancy:
        bar = "#" * 76
        logbuf += ["", bar, "### " + stamp + msg, bar, ""]
    else:
        logbuf.append(stamp + msg)
    if len(logbuf) > 10000 and not logfile:
        del logbuf[:5000]
    if logfile:
        with open(logfile, "a") as f:
            f.write("\n".join(logbuf) + "\n")
        logbuf = []
    if not quiet:
        print(msg)


def qlog(*args):
    log(*args, quiet=True)
