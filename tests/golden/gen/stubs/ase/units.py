"""Import-only stand-in: the two constants the reference's calculators read."""
GPa = 1.0 / 160.21766208
kB = 8.617330337217213e-05
