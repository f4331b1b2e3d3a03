import sys
import numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
dx = (a["x"].view(np.uint32) != b["x"].view(np.uint32)).any(axis=1)
print(sys.argv[1], "vs", sys.argv[2], ": problems with different x bits", int(dx.sum()), "different iterations", int((a["it"] != b["it"]).sum()),
      "different status", int((a["st"] != b["st"]).sum()), "max |dx|", float(np.abs(a["x"] - b["x"]).max()))
