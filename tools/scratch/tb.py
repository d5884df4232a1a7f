import faulthandler, runpy, sys
faulthandler.dump_traceback_later(int(sys.argv[1]), exit=True)
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
