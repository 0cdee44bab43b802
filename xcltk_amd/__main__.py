from .xcltk import main

main()
