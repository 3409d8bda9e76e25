"""TEST INFRASTRUCTURE ONLY -- see gensim/models/__init__.py"""
