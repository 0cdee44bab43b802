"""Global identifiers. APP/VERSION mirror the reference CLI this package is a drop-in for
(xcltk/config.py:3-4); ENGINE is this implementation."""
APP = "xcltk"
VERSION = "0.5.2"
ENGINE = "xcltk_amd 0.1.0 (MI355X / gfx950 HIP engine)"
